"""Drop-in for the reference's serial_hex_to_excel_v3.py: hex capture text -> "Parsed" workbook.

Same module-level names, arguments, return values and CLI as the reference script
(serial_hex_to_excel_v3.py:18-36 _read_bytes, :41-103 _parse_groups, :105-120 _write_excel,
:122-158 main); the byte unpack and the resynchronising record scan run on the GPU through
libslamhip.so.  There is no CPU implementation behind these functions: without the library or a
GPU they raise.

Difference a caller can see: the workbook is written with the standard library (no openpyxl
needed).  The log lines are reproduced one by one: the device returns which tokens _read_bytes
skipped (index and byte span) and where the scan of _parse_groups discarded a group; the token
text, the reason and the hex dump of each line are read off the input at those places.
"""
from __future__ import annotations

import logging
import sys
from pathlib import Path
from typing import List, Tuple

import numpy as np

from slamhip import ops, xlsx

HEADER = ["FLAG", "UE_Beam[5:0]十进制", "BS_Beam[5:0]十进制", "RSS十进制", "CLK十进制"]   # :110-116


def _to_hex(v: int) -> str:
    return "0x%02X" % v


def _top2(v: int) -> int:
    return (v >> 6) & 3


def _lsb6(v: int) -> int:
    return v & 63


def _fmt_bytes(seq: List[int]) -> str:
    return " ".join(_to_hex(v) for v in seq)


def _read_bytes(path: Path, logger: logging.Logger) -> List[int]:
    """File -> list of byte values, token rules of the reference (:16, :18-36)."""
    raw = Path(path).read_bytes()
    data = ops.hex_unpack(raw)
    if logger.isEnabledFor(logging.DEBUG):
        for line in skipped_token_lines(raw):
            logger.debug(line)
    return data.tolist()


def skipped_token_lines(raw: bytes) -> List[str]:
    """The line the reference logs for each token it skips (:27): the device returns the token's index and
    byte span, the token text is that span decoded the way the file was (UTF-8, errors ignored)."""
    idx, beg, end = ops.token_rejects(raw)
    return ["跳过非16进制单字节token[%d]='%s'" % (int(i), bytes(raw[int(b):int(e)]).decode("utf-8", errors="ignore"))
            for i, b, e in zip(idx.tolist(), beg.tolist(), end.tolist())]


def _parse_groups(bytes_list: List[int], logger: logging.Logger) -> Tuple[List[Tuple[int, int, int, int, int]], int, int]:
    """Byte values -> (rows of (flag, ue, bs, rss, clk), valid, discarded) (:41-103)."""
    if len(bytes_list) and not isinstance(bytes_list, (bytes, bytearray)):
        arr = np.asarray(bytes_list)
        if arr.dtype.kind not in "iu" or arr.min() < 0 or arr.max() > 255:
            # _read_bytes only yields 0..255; anything else cannot be carried as device bytes without changing
            # which positions look like flag bytes
            raise ValueError("bytes_list must hold integers in 0..255")
    cols, valid, discarded = ops.parse_records(bytes_list)
    rows = list(zip(cols["flag"].tolist(), cols["ue"].tolist(), cols["bs"].tolist(),
                    cols["rss"].tolist(), cols["clk"].tolist()))
    if discarded:
        for line in reject_lines(bytes_list, ops.parse_rejects(bytes_list, discarded)):
            logger.error(line)
    return rows, valid, discarded


def reject_lines(bytes_list, positions) -> List[str]:
    """The "丢弃组" message the reference logs at each discard position (:55-57, :64, :69, :77, :85)."""
    n = len(bytes_list)
    out = []
    for i in (int(p) for p in positions):
        if i + 11 > n:
            out.append(f"丢弃组，起始索引{i}，原因：字节数量不足11个，剩余{n - i}字节，数据={_fmt_bytes(list(bytes_list[i:n]))}")
            continue
        grp = [int(v) for v in bytes_list[i:i + 11]]
        ue, bs = grp[1], grp[2]
        data = _fmt_bytes(grp)
        if _top2(ue) != 0b00:
            out.append(f"丢弃组，起始索引{i}，原因：UE_Beam位[7:6]不为00，UE={_to_hex(ue)}，数据={data}")
            continue
        if _top2(bs) != 0b11:
            out.append(f"丢弃组，起始索引{i}，原因：BS_Beam位[7:6]不为11，BS={_to_hex(bs)}，数据={data}")
            continue
        bad_clk = next((b for b in grp[3:8] if _top2(b) != 0b01), None)
        if bad_clk is not None:
            out.append(f"丢弃组，起始索引{i}，原因：CLK字节位[7:6]不为01，BAD={_to_hex(bad_clk)}，数据={data}")
            continue
        bad_rss = next((b for b in grp[8:11] if _top2(b) != 0b10), None)
        out.append(f"丢弃组，起始索引{i}，原因：RSS字节位[7:6]不为10，BAD={_to_hex(bad_rss if bad_rss is not None else 0)}，数据={data}")
    return out


def parse_file(path: Path, logger: logging.Logger):
    """Fused fast path: file -> (columns dict of NumPy arrays, valid, discarded) in one device call."""
    return ops.parse_text(Path(path).read_bytes())


def _write_excel(rows, out_path: Path) -> None:
    """Rows -> workbook with sheet "Parsed" and the reference's five headers (:105-120)."""
    out_path = Path(out_path)
    out_path.parent.mkdir(parents=True, exist_ok=True)
    cols = [list(c) for c in zip(*rows)] if rows else [[] for _ in HEADER]
    xlsx.write_table(str(out_path), HEADER, cols, sheet="Parsed")


def main() -> None:
    default_in = Path(r"D:\Code\SLAMPro\UE21\debugDoc\Serial Debug 2026-01-27 115303.txt")
    default_out = Path(r"D:\Code\SLAMPro\UE21\debugDoc\Serial Debug 2026-01-27 115303.xlsx")
    in_path = Path(sys.argv[1]) if len(sys.argv) >= 2 else default_in
    out_path = Path(sys.argv[2]) if len(sys.argv) >= 3 else default_out
    if not in_path.exists():
        print(f"输入文件不存在: {in_path}")
        return
    log_path = out_path.with_suffix(".log")
    logger = logging.getLogger("serial_hex_to_excel")
    logger.setLevel(logging.DEBUG)
    logger.handlers.clear()
    fh = logging.FileHandler(log_path.as_posix(), mode="w", encoding="utf-8")
    fh.setFormatter(logging.Formatter("%(asctime)s %(levelname)s %(message)s"))
    fh.setLevel(logging.DEBUG)
    sh = logging.StreamHandler()
    sh.setFormatter(logging.Formatter("%(levelname)s %(message)s"))
    sh.setLevel(logging.INFO)
    logger.addHandler(fh)
    logger.addHandler(sh)

    bytes_list = _read_bytes(in_path, logger)
    rows, valid, discarded = _parse_groups(bytes_list, logger)
    _write_excel(rows, out_path)
    summary = f"有效组数={valid} 丢弃组数={discarded} 输出={out_path} 日志={log_path}"
    logger.info(summary)
    print(summary)


if __name__ == "__main__":
    main()
