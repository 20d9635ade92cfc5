"""Minimal .xlsx reader/writer on the standard library (zipfile + ElementTree).

The reference moves data between its three stages as workbooks written by
openpyxl (serial_hex_to_excel_v3.py:105-120) or pandas
(bs_beam_correction.py:202-203) and sometimes re-saved by Excel.  This module
reads those three cell encodings -- inline strings, shared strings, bare
numbers -- and writes a single-sheet workbook any of those tools can open, so
the drop-in scripts do not need openpyxl.
"""
from __future__ import annotations

import re
import zipfile
from typing import Dict, Iterable, List, Optional, Sequence
from xml.etree import ElementTree as ET
from xml.sax.saxutils import escape

_NS = "{http://schemas.openxmlformats.org/spreadsheetml/2006/main}"
_RNS = "{http://schemas.openxmlformats.org/officeDocument/2006/relationships}"
_PKG_RNS = "{http://schemas.openxmlformats.org/package/2006/relationships}"
_REF = re.compile(r"([A-Z]+)(\d+)")


def _col_index(letters: str) -> int:
    n = 0
    for ch in letters:
        n = n * 26 + (ord(ch) - 64)
    return n - 1


def _col_letters(idx: int) -> str:
    s = ""
    idx += 1
    while idx:
        idx, r = divmod(idx - 1, 26)
        s = chr(65 + r) + s
    return s


def _num(text: str):
    try:
        return int(text)
    except ValueError:
        f = float(text)
        return int(f) if f.is_integer() and "e" not in text.lower() and "." not in text else f


def sheet_names(path: str) -> List[str]:
    with zipfile.ZipFile(path) as z:
        wb = ET.fromstring(z.read("xl/workbook.xml"))
        return [s.get("name") for s in wb.iter(_NS + "sheet")]


def _sheet_targets(z: zipfile.ZipFile) -> Dict[str, str]:
    wb = ET.fromstring(z.read("xl/workbook.xml"))
    rels = ET.fromstring(z.read("xl/_rels/workbook.xml.rels"))
    rid_to_target = {r.get("Id"): r.get("Target") for r in rels.iter(_PKG_RNS + "Relationship")}
    out = {}
    for s in wb.iter(_NS + "sheet"):
        tgt = rid_to_target[s.get(_RNS + "id")]
        tgt = tgt.lstrip("/")
        if not tgt.startswith("xl/"):
            tgt = "xl/" + tgt
        out[s.get("name")] = tgt
    return out


def _shared_strings(z: zipfile.ZipFile) -> List[str]:
    if "xl/sharedStrings.xml" not in z.namelist():
        return []
    root = ET.fromstring(z.read("xl/sharedStrings.xml"))
    return ["".join(t.text or "" for t in si.iter(_NS + "t")) for si in root.iter(_NS + "si")]


def read_rows(path: str, sheet: Optional[str] = None) -> List[List[object]]:
    """All rows of one sheet (the first when `sheet` is None) as lists; empty cells are None."""
    with zipfile.ZipFile(path) as z:
        targets = _sheet_targets(z)
        if sheet is None:
            sheet = next(iter(targets))
        if sheet not in targets:
            raise KeyError(f"worksheet not found: {sheet}")
        sst = _shared_strings(z)
        rows: List[List[object]] = []
        with z.open(targets[sheet]) as fh:
            for _, el in ET.iterparse(fh):
                if el.tag != _NS + "row":
                    continue
                cur: List[object] = []
                for c in el.iter(_NS + "c"):
                    m = _REF.match(c.get("r") or "")
                    ci = _col_index(m.group(1)) if m else len(cur)
                    while len(cur) < ci:
                        cur.append(None)
                    t = c.get("t")
                    v = c.find(_NS + "v")
                    val: object = None
                    if t == "inlineStr":
                        val = "".join(x.text or "" for x in c.iter(_NS + "t"))
                    elif v is not None and v.text is not None:
                        if t == "s":
                            val = sst[int(v.text)]
                        elif t in ("str", "e"):
                            val = v.text
                        elif t == "b":
                            val = bool(int(v.text))
                        else:
                            val = _num(v.text)
                    if len(cur) == ci:
                        cur.append(val)
                    else:
                        cur[ci] = val
                rn = el.get("r")
                if rn is not None:
                    while len(rows) < int(rn) - 1:
                        rows.append([])
                rows.append(cur)
                el.clear()
        return rows


def read_table(path: str, sheet: Optional[str] = None) -> Dict[str, list]:
    """First row is the header; returns {column name: list of values} (short rows padded with None)."""
    rows = read_rows(path, sheet)
    if not rows:
        return {}
    header = [h for h in rows[0]]
    cols: Dict[str, list] = {}
    width = len(header)
    for j, h in enumerate(header):
        if h is None:
            continue
        cols[str(h)] = [r[j] if j < len(r) else None for r in rows[1:]]
    _ = width
    return cols


def read_all_tables(path: str) -> Dict[str, Dict[str, list]]:
    return {name: read_table(path, name) for name in sheet_names(path)}


_CT = """<?xml version="1.0" encoding="UTF-8" standalone="yes"?>
<Types xmlns="http://schemas.openxmlformats.org/package/2006/content-types"><Default Extension="rels" ContentType="application/vnd.openxmlformats-package.relationships+xml"/><Default Extension="xml" ContentType="application/xml"/><Override PartName="/xl/workbook.xml" ContentType="application/vnd.openxmlformats-officedocument.spreadsheetml.sheet.main+xml"/><Override PartName="/xl/worksheets/sheet1.xml" ContentType="application/vnd.openxmlformats-officedocument.spreadsheetml.worksheet+xml"/></Types>"""
_RELS = """<?xml version="1.0" encoding="UTF-8" standalone="yes"?>
<Relationships xmlns="http://schemas.openxmlformats.org/package/2006/relationships"><Relationship Id="rId1" Type="http://schemas.openxmlformats.org/officeDocument/2006/relationships/officeDocument" Target="xl/workbook.xml"/></Relationships>"""
_WB = """<?xml version="1.0" encoding="UTF-8" standalone="yes"?>
<workbook xmlns="http://schemas.openxmlformats.org/spreadsheetml/2006/main" xmlns:r="http://schemas.openxmlformats.org/officeDocument/2006/relationships"><sheets><sheet name="{name}" sheetId="1" r:id="rId1"/></sheets></workbook>"""
_WBRELS = """<?xml version="1.0" encoding="UTF-8" standalone="yes"?>
<Relationships xmlns="http://schemas.openxmlformats.org/package/2006/relationships"><Relationship Id="rId1" Type="http://schemas.openxmlformats.org/officeDocument/2006/relationships/worksheet" Target="worksheets/sheet1.xml"/></Relationships>"""


def write_table(path: str, header: Sequence[str], columns: Sequence[Iterable], sheet: str = "Sheet1") -> None:
    """One sheet: a header row of inline strings, then numeric rows (None -> empty cell)."""
    cols = [list(c) for c in columns]
    nrows = len(cols[0]) if cols else 0
    letters = [_col_letters(j) for j in range(len(header))]
    parts = ['<?xml version="1.0" encoding="UTF-8" standalone="yes"?>\n'
             '<worksheet xmlns="http://schemas.openxmlformats.org/spreadsheetml/2006/main"><sheetData>']
    parts.append('<row r="1">' + "".join(
        f'<c r="{letters[j]}1" t="inlineStr"><is><t>{escape(str(h))}</t></is></c>' for j, h in enumerate(header)
    ) + "</row>")
    for i in range(nrows):
        r = i + 2
        cells = []
        for j, col in enumerate(cols):
            v = col[i]
            if v is None:
                continue
            if isinstance(v, str):
                cells.append(f'<c r="{letters[j]}{r}" t="inlineStr"><is><t>{escape(v)}</t></is></c>')
            else:
                if isinstance(v, float) and v != v:
                    continue
                cells.append(f'<c r="{letters[j]}{r}"><v>{v!r}</v></c>' if isinstance(v, float)
                             else f'<c r="{letters[j]}{r}"><v>{int(v)}</v></c>')
        parts.append(f'<row r="{r}">' + "".join(cells) + "</row>")
    parts.append("</sheetData></worksheet>")
    with zipfile.ZipFile(path, "w", zipfile.ZIP_DEFLATED) as z:
        z.writestr("[Content_Types].xml", _CT)
        z.writestr("_rels/.rels", _RELS)
        z.writestr("xl/workbook.xml", _WB.format(name=escape(sheet, {'"': "&quot;"})))
        z.writestr("xl/_rels/workbook.xml.rels", _WBRELS)
        z.writestr("xl/worksheets/sheet1.xml", "".join(parts))
