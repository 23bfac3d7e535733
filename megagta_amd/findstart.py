"""`findstart` on the device: seed k-mers for the HMM-guided search.

Host side of `megagta findstart <ref_aligned.faa> <reads.lib.bin> <k> [threads] [contigs.fa]` (fast_kmer_filter.cpp:49-190):
the reference word set (ProtKmerGenerator in model-only mode, prot_kmer_generator.h:60-135; first insertion wins, :88), the
call into mgta_findstart() for the scan of the reads, and the output lines (unique by nucleotide k-mer, :181-188).
The reference shuffles its lines (`random_shuffle`); here they come out sorted.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from ._lib import MegaGtaError, check

AA = "ARNDCQEGHILKMFPSTWYV"                       # prot_kmer.h:31-40 -> codes 0..19, '*' = 20
_CODE = {c: i for i, c in enumerate(AA)}
_DECODE = AA.lower() + "*"                        # ProtKmer::intToChar: lower case (prot_kmer.h:36-37,43)


class SeedHit(C.Structure):
    _fields_ = [("read", C.c_uint64), ("pos_strand", C.c_uint32), ("ref", C.c_int32)]


def read_fasta(path: str):
    name, chunks = None, []
    with open(path) as f:
        for line in f:
            line = line.rstrip("\r\n")
            if line.startswith(">"):
                if name is not None:
                    yield name, "".join(chunks)
                name, chunks = line[1:], []
            elif name is not None:
                chunks.append(line.strip())
    if name is not None:
        yield name, "".join(chunks)


def reference_words(faa_path: str, kaa: int) -> tuple[list[tuple[int, ...]], list[int]]:
    """words (tuples of residue codes) in insertion order + their model positions.  A window is broken by lower case (insert
    columns), '-' and 'X' ('-' and 'X' still occupy a model column); '.', '*' and letters outside the alphabet are skipped."""
    words: dict[tuple[int, ...], int] = {}
    for _, seq in read_fasta(faa_path):
        position, run, window = 1, 0, []
        for base in seq:
            if base.islower() or base in "-X":
                if base in "-X":
                    position += 1
                run = 0
                continue
            if base in _CODE:
                window.append(_CODE[base])
                position += 1
                run += 1
                if run >= kaa:
                    words.setdefault(tuple(window[-kaa:]), position - kaa)
    return list(words.keys()), list(words.values())


def pack_words(words: list[tuple[int, ...]], kaa: int) -> np.ndarray:
    """[n][2] uint64: first min(12, kaa) residues | the rest, 5 bits each, first residue highest (kmer.h:66-84)"""
    out = np.zeros((len(words), 2), dtype=np.uint64)
    for i, w in enumerate(words):
        w0 = w1 = 0
        for j, c in enumerate(w):
            if j < 12:
                w0 = (w0 << 5) | c
            else:
                w1 = (w1 << 5) | c
        out[i, 0], out[i, 1] = w0, w1
    return out


def find_hits(ctx, reads, reads_reversed: bool, k: int, packed_words: np.ndarray) -> tuple[np.ndarray, float]:
    """raw hits of mgta_findstart as a structured array (read, pos_strand, ref) + kernel milliseconds"""
    pw = np.ascontiguousarray(packed_words, dtype=np.uint64)
    cap = 1 << 16
    while True:
        hits = (SeedHit * cap)()
        n, ms = C.c_int64(0), C.c_double(0)
        check(ctx._L.mgta_findstart(ctx.h, reads.h, int(reads_reversed), k, pw.ctypes.data, pw.shape[0], hits, cap, C.byref(n), C.byref(ms)),
              "mgta_findstart")
        if n.value <= cap:
            arr = np.frombuffer(hits, dtype=[("read", "<u8"), ("pos_strand", "<u4"), ("ref", "<i4")], count=n.value).copy()
            return arr, ms.value
        cap = int(n.value) + 1024


_COMP = bytes.maketrans(b"ACGT", b"TGCA")


def seed_lines(hits: np.ndarray, read_string, words: list[tuple[int, ...]], model_pos: list[int], k: int) -> list[str]:
    """`read_string(i)` -> the read as sequenced (ACGT str).  Lines of `megagta findstart`, unique by k-mer, sorted."""
    seen: dict[str, tuple[str, int]] = {}
    cache: dict[int, tuple[str, str]] = {}
    for r, ps, ref in zip(hits["read"].tolist(), hits["pos_strand"].tolist(), hits["ref"].tolist()):
        if r not in cache:
            s = read_string(r)
            cache[r] = (s, s.encode().translate(_COMP)[::-1].decode())
        s = cache[r][ps & 1]
        nucl = s[ps >> 1:(ps >> 1) + k]
        if nucl not in seen:
            seen[nucl] = ("".join(_DECODE[c] for c in words[ref]), model_pos[ref])
    return sorted(f"dump_gene_name\tdump_seq_name\tdump\t{n}\ttrue\t1\t{p}\t{m}" for n, (p, m) in seen.items())


def find_start(ctx, faa_path: str, read_codes: list[np.ndarray], k: int, contig_codes: list[np.ndarray] = ()) -> tuple[list[str], dict]:
    """whole step for in-process callers: reads (+ contigs of the previous k) as arrays of base codes, forward orientation"""
    from . import readlib
    if k % 3 != 0 or not 9 <= k <= 72:
        raise MegaGtaError(f"findstart: k = {k}: a multiple of 3 in [9, 72] is required")
    seqs = list(read_codes) + list(contig_codes)
    words, mpos = reference_words(faa_path, k // 3)
    packed, start = readlib.pack_for_build(seqs)          # reversed storage: the same upload serves buildgraph
    rd = ctx.upload_reads(packed, start)
    try:
        hits, ms = find_hits(ctx, rd, True, k, pack_words(words, k // 3))
    finally:
        rd.free()
    lines = seed_lines(hits, lambda i: "".join("ACGT"[x] for x in seqs[i]), words, mpos, k)
    return lines, {"n_hits": int(hits.size), "n_seeds": len(lines), "n_ref_words": len(words), "ms_kernel": ms}
