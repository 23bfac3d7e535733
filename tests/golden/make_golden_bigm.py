#!/usr/bin/env python3
"""Goldens for profile HMMs that do NOT fit the LDS of a CU -- (M + 1)(A + 11) * 8 B of tables beside the heap tops > 160 KB, i.e. every
model longer than ~400 columns (the reference has no bound on M: profile_hmm.h:11-100) -- and for HMMER3 text with `*` entries and a
non-uniform COMPO line (hmmer3b_parser.h:63-75,122-172).  Everything comes from the COMPILED REFERENCE (oracle/_ref/megagta + probe):

    python tests/golden/make_golden_bigm.py

Inputs are regenerated from their seeds by the tests (megagta_amd.synth is deterministic; an md5 of every input is stored and checked),
so only the reference's answers are committed: seeds from its `findstart`, per-seed cold and sequential warm A* results from
`probe astar`, parsed tables from `probe hmm`."""
from __future__ import annotations

import hashlib
import json
import os
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from megagta_amd import synth          # noqa: E402
from tests.golden.make_golden import run, gz_write, buildgraph, REF, GOLD   # noqa: E402

from tests.helpers import BIGM_CASES as CASES, bigm_inputs   # noqa: E402  (the tests regenerate the same inputs)


def star_hmm_text() -> str:
    """a 12-column model with `*` match emissions and `*` transitions, a non-uniform COMPO line and a lower-case alphabet letter"""
    import numpy as np
    rng = np.random.default_rng(77)
    letters = list("ACDEFGHIKLMNPQRSTVWY")
    letters[3] = "e"                                     # parseAlpha maps either case (hmmer3b_parser.h:179-201)
    M = 12
    out = ["HMMER3/b [synthetic]", "NAME  star", f"LENG  {M}", "ALPH  amino", "HMM     " + "   ".join(letters),
           "        m->m     m->i     m->d     i->m     i->i     d->m     d->d"]
    compo = rng.dirichlet(np.ones(20) * 3)
    out.append("  COMPO " + " ".join(f"{-np.log(p):.5f}" for p in compo))
    ins = " ".join(f"{-np.log(p):.5f}" for p in np.full(20, 0.05))
    out.append("        " + ins)
    out.append("        " + " ".join(["0.01005", "5.29832", "5.29832", "0.61958", "0.77255", "0.00000", "*"]))
    for i in range(1, M + 1):
        p = rng.dirichlet(np.ones(20))
        toks = [f"{-np.log(x):.5f}" for x in p]
        for j in rng.choice(20, size=3 if i % 3 else 0, replace=False):
            toks[int(j)] = "*"
        out.append(f"{i:7d} " + " ".join(toks))
        out.append("        " + ins)
        tr = ["0.03046", "4.60517", "4.19971", "0.61958", "0.77255", "0.35667", "1.20397"]
        if i % 4 == 0:
            tr[2] = "*"                                  # no m->d here
        if i % 5 == 0:
            tr[1] = "*"; tr[4] = "*"                     # no inserts here
        if i == M:
            tr = ["0.00000", "*", "*", "0.00000", "*", "0.00000", "*"]
        out.append("        " + " ".join(tr))
    out.append("//")
    return "\n".join(out) + "\n"


def main():
    out = os.path.join(GOLD, "bigm")
    shutil.rmtree(out, ignore_errors=True)
    os.makedirs(out)
    tmp = tempfile.mkdtemp(prefix="mgta_gold_bigm_")
    meta = {}
    for case, c in CASES.items():
        d = os.path.join(tmp, case)
        os.makedirs(d)
        mg, gdir = bigm_inputs(case, d)
        synth.write_fasta(mg.reads, f"{d}/reads.fa")
        open(f"{d}/reads.lib", "w").write(f"reads.fa\nse {d}/reads.fa\n")
        run([f"{REF}/megagta", "buildlib", f"{d}/reads.lib", f"{d}/reads.lib"])
        buildgraph(f"{d}/reads.lib", f"{d}/k44/44", 44)
        seeds = run([f"{REF}/megagta", "findstart", f"{gdir}/ref_aligned.faa", f"{d}/reads.lib.bin", "45", "1"]).stdout.splitlines()
        seeds = sorted(seeds)                              # (the reference shuffles its lines: fast_kmer_filter.cpp:183)
        step = max(1, len(seeds) // c["n_seeds"])
        seeds = seeds[::step][:c["n_seeds"]]
        open(f"{out}/{case}_starting_kmers.txt", "wb").write(b"\n".join(seeds) + b"\n")
        for mode in ("cold", "warm"):
            res = run([f"{REF}/probe", "astar", f"{d}/k44/44", f"{gdir}/for_enone.hmm", f"{gdir}/rev_enone.hmm",
                       f"{out}/{case}_starting_kmers.txt", "20", "0.5", mode]).stdout
            gz_write(f"{out}/{case}_astar_{mode}.txt.gz", res)
        meta[case] = dict(c, n_seeds_found=len(seeds),
                          md5={n: hashlib.md5(open(p, "rb").read()).hexdigest()
                               for n, p in (("reads", f"{d}/reads.lib.bin"), ("for", f"{gdir}/for_enone.hmm"), ("rev", f"{gdir}/rev_enone.hmm"))})
    open(f"{out}/star.hmm", "w").write(star_hmm_text())
    gz_write(f"{out}/star_tables.txt.gz", run([f"{REF}/probe", "hmm", f"{out}/star.hmm"]).stdout)
    json.dump(meta, open(f"{out}/cases.json", "w"), indent=1)
    shutil.rmtree(tmp)
    print("written", out, meta)


if __name__ == "__main__":
    main()
