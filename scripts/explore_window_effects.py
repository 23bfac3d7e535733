#!/usr/bin/env python3
"""Where does cache sharing change contigs?  For a few synthetic settings: cold vs window 1 (= `search ... 1`) vs window B in one batch
vs the seeds of the gene split over two "ranks" (two batches over the even / odd seeds, as search_dist.py runs them).  Prints, per
setting, how many contigs differ.  (Used to pick the input of tests/test_multi_gpu_gpu.py::test_split_gene_agreement.)"""
import os, sys, tempfile
from collections import Counter
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from megagta_amd import api, synth, hmm as hmmlib, findstart, dist as mdist


def run(ctx, n_reads, M, aa_sub, err, rpg, glen, prune, pen, window, seed):
    mg = synth.make_metagenome(n_reads, 150, (("g", M),), seed=seed, reads_per_genome=rpg, genome_len=glen, aa_sub=aa_sub, err=err)
    packed, start = synth.pack_reads_for_build(mg.reads)
    stream = ctx.build_sdbg(ctx.upload_reads(packed, start), 44)
    with tempfile.TemporaryDirectory() as td:
        synth.write_gene_models(mg.genes, td)
        f, r, faa = (os.path.join(td, "g", x) for x in ("for_enone.hmm", "rev_enone.hmm", "ref_aligned.faa"))
        lines, _ = findstart.find_start(ctx, faa, list(mg.reads), 45)
        seeds = [(l.split("\t")[3], int(l.split("\t")[7])) for l in lines]
        g = api.Graph(ctx, stream)
        fw, rv = api.DeviceHmm(ctx, hmmlib.parse_hmm(f)), api.DeviceHmm(ctx, hmmlib.parse_hmm(r))
        km, ss = [s[0] for s in seeds], [s[1] - 1 for s in seeds]

        def go(idx, mode):
            c, o, st = api.astar_search_packed(g, fw, rv, [km[i] for i in idx], [ss[i] for i in idx], prune, pen, cache_mode=mode)
            return mdist.contig_list(c, o), st["n_expansions"]
        allidx = list(range(len(km)))
        cold, e0 = go(allidx, 0)
        seq, e1 = go(allidx, 1)
        win, e2 = go(allidx, window)
        two = [None] * len(km)
        e3 = 0
        for r_ in range(2):
            idx = allidx[r_::2]
            res, e = go(idx, window)
            e3 += e
            for i, c in zip(idx, res):
                two[i] = c
        d = lambda a, b: sum(1 for x, y in zip(a, b) if x != y)
        ms = lambda a, b: sum((Counter(a) & Counter(b)).values())
        print(f"reads {n_reads} M {M} aa_sub {aa_sub} err {err} cov {rpg * 150 / glen:.0f}x prune {prune} pen {pen} window {window}: {len(km)} seeds | "
              f"differ from seq: cold {d(cold, seq)}, window {d(win, seq)}, two-rank {d(two, seq)} | two-rank vs one-rank window: {d(two, win)} "
              f"(multiset common {ms(two, win)}) | expansions {e0} {e1} {e2} {e3}", flush=True)
        g.free()


if __name__ == "__main__":
    ctx = api.Context(0)
    for cfg in [(12000, 277, 0.10, 0.005, 1000, 12000, 20, 0.5, 16, 23), (12000, 277, 0.10, 0.005, 1000, 12000, 20, 0.5, 4, 23),
                (20000, 277, 0.03, 0.01, 500, 12000, 20, 0.5, 8, 5), (20000, 277, 0.03, 0.02, 300, 12000, 10, 0.5, 8, 6),
                (20000, 200, 0.02, 0.02, 250, 10000, 5, 0.9, 8, 7), (30000, 277, 0.05, 0.015, 400, 12000, 20, 0.5, 8, 8),
                (30000, 120, 0.01, 0.03, 200, 8000, 3, 0.5, 8, 9)]:
        try:
            run(ctx, *cfg)
        except Exception as e:
            print("failed", cfg, e, flush=True)
