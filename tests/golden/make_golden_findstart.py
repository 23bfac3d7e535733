#!/usr/bin/env python3
"""Regenerates tests/golden/findstart/* from the COMPILED REFERENCE (`oracle/_ref/megagta findstart`, fast_kmer_filter.cpp).

    python tests/golden/make_golden_findstart.py        (build container only: needs oracle/_ref)

Inputs are seeded synthetic data: a reference alignment that exercises every rule of the model-only k-mer generator
(lower-case insert columns, '-', '.', 'X', '*', letters outside the alphabet, a sequence shorter than k, k-mers shared by two
sequences at different model positions), reads that carry back-translations of its k-mers on both strands, reads shorter than k,
reads with N, and a multi-line contig FASTA for the optional fifth argument.  Outputs: the reference's seed lines, SORTED (the
reference shuffles them).  Data only.
"""
from __future__ import annotations

import gzip
import os
import shutil
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
REF = os.path.join(ROOT, "oracle", "_ref")
OUT = os.path.join(ROOT, "tests", "golden", "findstart")

AA = "ARNDCQEGHILKMFPSTWYV"
CODON_AA = "KNKNTTTTRSRSIIMIQHQHPPPPRRRRLLLLEDEDAAAAGGGGVVVV*Y*YSSSS*CWCLFLF"


def run(cmd, **kw):
    return subprocess.run(cmd, check=True, stdout=subprocess.PIPE, stderr=subprocess.PIPE, **kw)


def main():
    rng = np.random.default_rng(2024)
    shutil.rmtree(OUT, ignore_errors=True)
    os.makedirs(OUT)
    tmp = tempfile.mkdtemp(prefix="mgta_fs_")
    codons = {a: [i for i in range(64) if CODON_AA[i] == a] for a in AA + "*"}
    base = "".join(AA[i] for i in rng.integers(0, 20, 140))
    s1 = base[:30] + "--" + base[30:52] + "acd" + base[52:80] + ".." + base[80:110] + "-" + base[110:]
    s2 = base[:12] + "X" + base[13:40] + "kk" + base[40:75] + "*" + base[75:100] + "BZU" + base[100:] + "".join(AA[i] for i in rng.integers(0, 20, 30))
    s3 = "".join(AA[i] for i in rng.integers(0, 20, 60)) + base[20:60].lower() + base[60:90] + "x" + base[90:120]
    s4 = "MKV"
    with open(f"{OUT}/ref_quirks.faa", "w") as f:
        for i, s in enumerate((s1, s2, s3, s4)):
            f.write(f">seq{i} some description\n")
            for j in range(0, len(s), 50):
                f.write(s[j:j + 50] + "\n")
    prot_sources = [base, "".join(c for c in s2 if c in AA), "".join(c for c in s3 if c in AA)]

    def backtranslate(p):
        out = []
        for a in p:
            c = int(rng.choice(codons[a]))
            out += [c >> 4, (c >> 2) & 3, c & 3]
        return out

    def embed(total):
        src = prot_sources[int(rng.integers(0, len(prot_sources)))]
        n_aa = int(rng.integers(8, 40))
        st = int(rng.integers(0, max(1, len(src) - n_aa)))
        core = backtranslate(src[st:st + n_aa])
        left = int(rng.integers(0, max(1, total - len(core) + 1))) if total > len(core) else 0
        seq = list(rng.integers(0, 4, left)) + core
        seq += list(rng.integers(0, 4, max(0, total - len(seq))))
        seq = np.array(seq[:max(total, 1)], dtype=np.uint8)
        if rng.random() < 0.5:
            seq = (3 - seq[::-1]).astype(np.uint8)
        return seq

    reads = []
    for _ in range(400):
        reads.append(embed(int(rng.integers(50, 220))))
    for _ in range(100):
        reads.append(rng.integers(0, 4, int(rng.integers(10, 200))).astype(np.uint8))
    with open(f"{tmp}/reads.fa", "w") as f:
        for i, r in enumerate(reads):
            s = "".join("ACGT"[x] for x in r)
            if i % 97 == 0 and len(s) > 20:
                s = s[:7] + "N" + s[8:15] + "n" + s[16:]
            f.write(f">r{i}\n{s}\n")
    open(f"{tmp}/reads.lib", "w").write(f"reads.fa\nse {tmp}/reads.fa\n")
    run([f"{REF}/megagta", "buildlib", f"{tmp}/reads.lib", f"{tmp}/reads.lib"])
    shutil.copy(f"{tmp}/reads.lib.bin", f"{OUT}/reads.lib.bin")
    open(f"{OUT}/reads.lib.lib_info", "w").write(open(f"{tmp}/reads.lib.lib_info").read())
    with open(f"{OUT}/contigs.fa", "w") as f:
        for i in range(6):
            parts = [embed(int(rng.integers(100, 400))) for _ in range(int(rng.integers(2, 6)))]
            s = "".join("ACGT"[x] for x in np.concatenate(parts))
            f.write(f">k29_{i} flag=1 multi=3.0000 len={len(s)}\n")
            for j in range(0, len(s), 70):
                f.write(s[j:j + 70] + "\n")
    for k in (30, 45, 72):
        for with_contigs in (False, True):
            cmd = [f"{REF}/megagta", "findstart", f"{OUT}/ref_quirks.faa", f"{OUT}/reads.lib.bin", str(k), "2"]
            if with_contigs:
                cmd.append(f"{OUT}/contigs.fa")
            lines = sorted(run(cmd).stdout.decode().splitlines())
            name = f"{OUT}/seeds_k{k}{'_contigs' if with_contigs else ''}.txt.gz"
            with gzip.GzipFile(name, "wb", mtime=0) as g:
                g.write("".join(l + "\n" for l in lines).encode())
            print(name, len(lines))
    shutil.rmtree(tmp, ignore_errors=True)


if __name__ == "__main__":
    main()
