"""Deterministic synthetic capture generator (NumPy twin of csrc/k_synth.hip: byte-identical output).

A frame of shape B x N is the hex text of B UE sweeps ("rows") of N records each, in the
format of the reference's debugDoc captures: "« " then "XX " per byte, no newlines.

Per record r = b*N + n of frame `frame`:  h = splitmix64(seed ^ (frame << 40) ^ r)
  ue    = (n*64)//N                        a 0..63 ramp per row, so every row is one group
  n % 64 == 63  -> handover record: FLAG byte 0xCC, bs = (b + n//64) % 64, rss copied from record
                   n-1 (that pair makes a baseline), clock on the cycle grid
  otherwise     -> FLAG byte 0x33, bs = the previous handover's bs (63 before the first),
                   rss = 15000 + (h>>16) % 78000
  clk   = 1563 + n*61000 + jitter;  jitter = (h % 401) - 200, except that 40 % of the plain records
          ((h>>40) % 100 < 40, never the record before a handover) sit off the grid by
          2000 + (h>>8) % 57000 and are therefore dropped by the correction
  one record in 256 ((h>>52) % 256 == 0) gets bit 6 of byte (h>>60) % 11 flipped
Every row starts with 7 junk bytes 33 33 33 33 33 2F E2 (padding + a truncated record).
"""
from __future__ import annotations

import hashlib

import numpy as np

BASE_SEED = 20260126
JUNK = np.array([0x33, 0x33, 0x33, 0x33, 0x33, 0x2F, 0xE2], np.uint8)
_M = np.uint64(0xFFFFFFFFFFFFFFFF)


def splitmix64(x: np.ndarray) -> np.ndarray:
    x = (x.astype(np.uint64) + np.uint64(0x9E3779B97F4A7C15)) & _M
    z = x
    z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _M
    z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _M
    return z ^ (z >> np.uint64(31))


def text_len(B: int, N: int) -> int:
    return 3 + 3 * (B * (len(JUNK) + 11 * N))


def make_frame_bytes(B: int, N: int, frame: int, seed: int = BASE_SEED) -> np.ndarray:
    """The data bytes of one frame (junk included), uint8 [B*(7+11N)]."""
    with np.errstate(over="ignore"):
        r = np.arange(B * N, dtype=np.uint64)
        b = (r // np.uint64(N)).astype(np.int64)
        n = (r % np.uint64(N)).astype(np.int64)
        key = np.uint64(seed) ^ (np.uint64(frame) << np.uint64(40))
        h = splitmix64(key ^ r)
        hprev = splitmix64(key ^ (r - np.uint64(1)))
    handover = (n % 64) == 63
    ue = (n * 64) // N
    bs = np.where(handover, (b + n // 64) % 64,
                  np.where(n < 63, 63, (b + (((n + 1) // 64) * 64 - 1) // 64) % 64))
    rss = np.where(handover, 15000 + ((hprev >> np.uint64(16)) % np.uint64(78000)).astype(np.int64),
                   15000 + ((h >> np.uint64(16)) % np.uint64(78000)).astype(np.int64))
    base = 1563 + n * 61000
    off = (((h >> np.uint64(40)) % np.uint64(100)).astype(np.int64) < 40) & ~handover & ((n % 64) != 62)
    clk = np.where(off, base + 2000 + ((h >> np.uint64(8)) % np.uint64(57000)).astype(np.int64),
                   base + (h % np.uint64(401)).astype(np.int64) - 200)
    rec = np.empty((B * N, 11), np.uint8)
    rec[:, 0] = np.where(handover, 0xCC, 0x33)
    rec[:, 1] = ue
    rec[:, 2] = 0xC0 | bs
    for j in range(5):
        rec[:, 3 + j] = 0x40 | ((clk >> (6 * j)) & 63)
    for j in range(3):
        rec[:, 8 + j] = 0x80 | ((rss >> (6 * j)) & 63)
    flip = ((h >> np.uint64(52)) % np.uint64(256)) == 0
    which = ((h >> np.uint64(60)) % np.uint64(11)).astype(np.int64)
    idx = np.flatnonzero(flip)
    rec[idx, which[idx]] ^= 0x40
    rows = np.empty((B, len(JUNK) + 11 * N), np.uint8)
    rows[:, : len(JUNK)] = JUNK
    rows[:, len(JUNK):] = rec.reshape(B, 11 * N)
    return rows.reshape(-1)


_HEX = np.frombuffer(b"0123456789ABCDEF", np.uint8)


def bytes_to_text(data: np.ndarray, prefix: bool = True) -> np.ndarray:
    """Capture format: optional '« ' then 'XX ' per byte (uint8 array)."""
    d = np.asarray(data, np.uint8)
    out = np.empty((len(d), 3), np.uint8)
    out[:, 0] = _HEX[d >> 4]
    out[:, 1] = _HEX[d & 15]
    out[:, 2] = 0x20
    body = out.reshape(-1)
    if not prefix:
        return body
    return np.concatenate([np.array([0xC2, 0xAB, 0x20], np.uint8), body])


def make_frame_text(B: int, N: int, frame: int, seed: int = BASE_SEED) -> np.ndarray:
    return bytes_to_text(make_frame_bytes(B, N, frame, seed))


def frame_digest(B: int, N: int, frame: int, seed: int = BASE_SEED) -> str:
    return hashlib.sha256(make_frame_text(B, N, frame, seed).tobytes()).hexdigest()
