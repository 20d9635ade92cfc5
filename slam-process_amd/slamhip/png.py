"""Minimal PNG writer for rendered RGBA tiles (standard library only: zlib + struct).

The device renders data-only tiles (no axes, no text); this writes one as an 8-bit RGBA PNG so that a batch
run can leave image files without going through matplotlib.  Rows are stored with filter type 0.
"""
from __future__ import annotations

import struct
import zlib

import numpy as np


def _chunk(tag: bytes, data: bytes) -> bytes:
    return struct.pack(">I", len(data)) + tag + data + struct.pack(">I", zlib.crc32(tag + data) & 0xFFFFFFFF)


def encode_rgba(tile: np.ndarray, level: int = 3) -> bytes:
    """[H, W, 4] uint8 -> PNG file bytes."""
    a = np.ascontiguousarray(tile, dtype=np.uint8)
    if a.ndim != 3 or a.shape[2] != 4:
        raise ValueError("tile must be [H, W, 4] uint8")
    h, w, _ = a.shape
    raw = np.zeros((h, 1 + w * 4), np.uint8)          # filter byte 0 in front of every row
    raw[:, 1:] = a.reshape(h, w * 4)
    return (b"\x89PNG\r\n\x1a\n" + _chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 6, 0, 0, 0)) +
            _chunk(b"IDAT", zlib.compress(raw.tobytes(), level)) + _chunk(b"IEND", b""))


def write_rgba(path: str, tile: np.ndarray, level: int = 3) -> None:
    with open(path, "wb") as fh:
        fh.write(encode_rgba(tile, level))
