"""Read-library host logic: `reads.lib.bin` / `.lib_info` <-> the packed, REVERSED read array the
SdBG build consumes.

Formats (reference): per read `uint32 len` + ceil(len/16) 2-bit words, forward orientation
(sequence_manager.cpp:375-410); `.lib_info` = "total_bases num_reads" + 2 lines per library
(read_lib_functions-inl.h:216-225).  `buildgraph` loads every read reversed, not complemented
(cx1_read2sdbg_s1.cpp:97,117) into one contiguous 2-bit array with base j of a word at bits 30-2j
(sequence_package.h:126-129).
"""
from __future__ import annotations

import numpy as np

_SHIFTS = (30 - 2 * np.arange(16)).astype(np.uint32)


def read_lib_info(prefix: str) -> tuple[int, int]:
    with open(prefix + ".lib_info") as f:
        total_bases, num_reads = (int(x) for x in f.readline().split()[:2])
    return total_bases, num_reads


def load_lib_bin(prefix: str) -> list[np.ndarray]:
    """Returns the reads as arrays of base codes (forward orientation)."""
    raw = np.fromfile(prefix + ".bin", dtype=np.uint32)
    _, num_reads = read_lib_info(prefix)
    reads, pos = [], 0
    for _ in range(num_reads):
        ln = int(raw[pos])
        nw = (ln + 15) // 16
        words = raw[pos + 1:pos + 1 + nw]
        codes = ((words[:, None] >> _SHIFTS[None, :]) & 3).astype(np.uint8).reshape(-1)[:ln]
        reads.append(codes)
        pos += 1 + nw
    if pos != raw.size:
        raise ValueError(f"{prefix}.bin: trailing data ({raw.size - pos} words)")
    return reads


def pack_codes(codes: np.ndarray) -> np.ndarray:
    n = codes.size
    nw = (n + 15) // 16
    buf = np.zeros(nw * 16, dtype=np.uint32)
    buf[:n] = codes
    return np.bitwise_or.reduce(buf.reshape(nw, 16) << _SHIFTS[None, :], axis=1).astype(np.uint32)


def pack_for_build(reads: list[np.ndarray]) -> tuple[np.ndarray, np.ndarray]:
    """(packed uint32 words, start_idx uint64[n+1]) with every read reversed."""
    lens = np.array([r.size for r in reads], dtype=np.uint64)
    start = np.zeros(len(reads) + 1, dtype=np.uint64)
    np.cumsum(lens, out=start[1:])
    flat = np.concatenate([r[::-1] for r in reads]) if reads else np.zeros(0, dtype=np.uint8)
    packed = pack_codes(flat)
    if packed.size == 0:
        packed = np.zeros(1, dtype=np.uint32)
    return packed, start


def load_for_build(prefix: str) -> tuple[np.ndarray, np.ndarray]:
    return pack_for_build(load_lib_bin(prefix))
