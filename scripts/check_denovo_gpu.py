"""device denovo vs the oracle on strain mixes: python scripts/check_denovo_gpu.py [n_cases]"""
import sys, time
sys.path.insert(0, ".")
import numpy as np
import torch  # noqa: F401
from megagta_amd import api, synth, readlib
from oracle import oracle

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 6
ctx = api.Context(0)
bad = 0
for case in range(n_cases):
    rng = np.random.default_rng(100 + case)
    k = int(rng.choice([15, 17, 21, 29, 31, 32, 33, 44, 63, 64, 65, 95, 127]))
    mc = int(rng.choice([1, 2, 3]))
    reads = synth.make_strain_mix(case, n_genomes=int(rng.integers(2, 7)), genome_len=int(rng.integers(1200, 5000)), read_len=int(max(100, k + 40)),
                                  snp_every=int(rng.choice([15, 25, 40, 80, 150])), cov=int(rng.choice([10, 20, 40])), err=float(rng.choice([0.0, 0.004, 0.01])),
                                  tricky=bool(rng.random() < 0.6))
    packed, start = readlib.pack_for_build(reads)
    ost = oracle.Stream.build(packed, start, k, threads=4) if mc == 1 else oracle.Stream.build_solid(packed, start, k, mc, False, threads=4)
    og = oracle.Graph(ost)
    for opts in ((150, False, k + 2), (-1, False, 0), (150, True, 0), (0, False, 0)):
        og = oracle.Graph(ost)
        t0 = time.time()
        want, wst = og.denovo(*opts)
        t1 = time.time()
        g = api.Graph(ctx, ost.edges())
        got, st = g.denovo(*opts)
        ok = got == want
        bad += not ok
        if ok and "-q" in sys.argv:
            continue
        print(f"case {case} k={k} m={mc} opts={opts}: edges {og.size}, oracle {wst} ({t1 - t0:.2f}s) | device tips {st['n_tips']} bubbles {st['n_bubbles']}"
              f" cand {st['n_bubble_candidates']} rounds {st['n_bubble_rounds']} paths {st['n_paths']} sweeps {st['n_unitig_sweeps']} contigs {st['n_contigs']}"
              f" ms {st['ms_tips']:.1f}/{st['ms_bubbles']:.1f}/{st['ms_unitigs']:.1f} -> {'OK' if ok else 'DIFF'}", flush=True)
        if not ok:
            a, b = want.splitlines(), got.splitlines()
            print("   lines", len(a), len(b), "same sorted seqs:", sorted(a[1::2]) == sorted(b[1::2]))
print("FAILED" if bad else "ALL OK", bad)
