"""Golden per-token log lines of the reference's _read_bytes (serial_hex_to_excel_v3.py:18-36).

Run in the build container:  python tests/golden/make_token_lines.py
Writes tests/golden/token_lines.json: {name: {"text": [byte values], "lines": [...], "n_bytes": kept tokens}}.
Inputs: the texts of edge_streams.json plus seeded mixes of good tokens, near-misses, multi-byte characters,
Unicode whitespace and invalid UTF-8; each is written to a file and read by the reference's own function.
"""
import json
import logging
import os
import sys
import tempfile
from pathlib import Path

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, "/root/reference")
import serial_hex_to_excel_v3 as S  # noqa: E402


class Capture(logging.Handler):
    def __init__(self):
        super().__init__(logging.DEBUG)
        self.lines = []

    def emit(self, record):
        self.lines.append(record.getMessage())


VOCAB = [b"33", b"CC", b"0x3f", b"0X3F", b"3", b"333", b"GG", b"x3", b"0x", b"0x123", b"fF", b"\xc2\xab", b"\xef\xbb\xbf33",
         b"\xe4\xb8\xad\xe6\x96\x87", b"\xf0\x9f\x98\x80", b"3\xff3", b"\xff", b"\xc3", b"\xe2\x82", b"\xed\xa0\x80", b"\xc0\xaf",
         b"\xf4\x90\x80\x80", b"4\x801", b"\x00", b"a\x00b", b"0x\xc2\xa0", b"1\xe2\x80\x8b2", b"'", b"%s", b"\\n"]
SPACES = [b" ", b"\n", b"\t", b"\r\n", b"\x0b", b"\x0c", b"\x1c", b"\x1f", b"\xc2\x85", b"\xc2\xa0", b"\xe3\x80\x80",
          b"\xe2\x80\x83", b"\xe2\x80\xa8", b"  ", b" \xe1\x9a\x80 "]


def texts():
    out = {}
    edge = json.load(open(os.path.join(HERE, "edge_streams.json")))
    for k, e in edge.items():
        out["edge_" + k] = bytes(e["text"])
    rng = np.random.default_rng(977)
    for j in range(24):
        parts = []
        for _ in range(int(rng.integers(1, 400))):
            parts.append(VOCAB[int(rng.integers(0, len(VOCAB)))] if rng.random() < 0.5 else b"%02X" % int(rng.integers(0, 256)))
            parts.append(SPACES[int(rng.integers(0, len(SPACES)))] if rng.random() < 0.9 else b"")
        out[f"mix_{j:02d}"] = b"".join(parts)
    out["long_garbage_token"] = b"33 " + b"z" * 9000 + b" CC " + b"\xe4\xb8\xad" * 3000 + b"\n44"
    out["only_invalid_bytes"] = bytes(rng.integers(0x80, 0x100, 500, dtype=np.uint8))
    out["random_bytes"] = bytes(rng.integers(0, 256, 6000, dtype=np.uint8))
    return out


def main():
    lg = logging.getLogger("golden_tokens")
    lg.setLevel(logging.DEBUG)
    lg.propagate = False
    doc = {}
    with tempfile.TemporaryDirectory() as d:
        for name, raw in texts().items():
            p = Path(d) / "t.txt"
            p.write_bytes(raw)
            cap = Capture()
            lg.handlers[:] = [cap]
            kept = S._read_bytes(p, lg)
            doc[name] = dict(text=list(raw), lines=cap.lines, n_bytes=len(kept))
    with open(os.path.join(HERE, "token_lines.json"), "w", encoding="utf-8") as fh:
        json.dump(doc, fh, ensure_ascii=False)
    print({k: (len(e["text"]), len(e["lines"])) for k, e in doc.items()})


if __name__ == "__main__":
    main()
