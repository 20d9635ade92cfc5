"""Golden per-reject log lines of the reference's _parse_groups (serial_hex_to_excel_v3.py:41-103).

Run in the build container (the reference is importable there):  python tests/golden/make_reject_lines.py
Writes tests/golden/reject_lines.json: {name: {"bytes": [...], "lines": [...], "valid": v, "discarded": d}}.
Inputs: the byte streams of edge_streams.json plus a few seeded streams dense in flag bytes and truncated tails.
"""
import json
import logging
import os
import sys

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


def rec(fl, ue, bs, clk, rss):
    return ([0xCC if fl else 0x33, ue & 63, 0xC0 | (bs & 63)] + [0x40 | ((clk >> (6 * j)) & 63) for j in range(5)] +
            [0x80 | ((rss >> (6 * j)) & 63) for j in range(3)])


def streams():
    out = {}
    edge = json.load(open(os.path.join(HERE, "edge_streams.json")))
    for k, e in edge.items():
        out["edge_" + k] = list(e["bytes"])
    rng = np.random.default_rng(4242)
    good = []
    for i in range(400):
        good += rec(i & 1, 0x33, 0x0C, 1563 + 61000 * i, 20000 + i)
    for cut in (11 * 7 + 1, 11 * 7 + 5, 11 * 7 + 10, 3072 + 3, 3072 - 4, 3024 + 7):
        out[f"cut_{cut}"] = good[:cut]
    out["flags_only"] = [0x33] * 200 + [0xCC] * 150
    out["noise"] = rng.integers(0, 256, 7000, dtype=np.uint8).tolist()
    out["tagged_noise"] = rng.choice(np.array([0x33, 0xCC, 0x00, 0x3F, 0xC0, 0xFF, 0x40, 0x7F, 0x80, 0xBF], np.uint8), 6000).tolist()
    flipped = list(good)
    for i in rng.integers(0, len(flipped), 120):
        flipped[int(i)] ^= 1 << int(rng.integers(0, 8))
    out["bit_flips"] = flipped
    return out


def main():
    lg = logging.getLogger("golden_rejects")
    lg.setLevel(logging.DEBUG)
    lg.propagate = False
    doc = {}
    for name, by in streams().items():
        cap = Capture()
        lg.handlers[:] = [cap]
        rows, v, d = S._parse_groups(list(by), lg)
        assert len(cap.lines) == d
        doc[name] = dict(bytes=list(map(int, by)), lines=cap.lines, valid=v, discarded=d)
    with open(os.path.join(HERE, "reject_lines.json"), "w", encoding="utf-8") as fh:
        json.dump(doc, fh, ensure_ascii=False)
    print({k: (e["valid"], e["discarded"]) for k, e in doc.items()})


if __name__ == "__main__":
    main()
