#!/usr/bin/env python3
"""Ordered-commit window B and cost term R at seed counts beyond the sizes the per-gene table was tuned on (<= 414 k seeds): the 100 M-read
graph resident, findstart's seeds of the first `sample` reads of one gene (sorted, unique: what `megagta findstart` writes), the same batch
under several (B, R).  R > 0: a search that has run p expansions releases p / R seeds beyond the window; R < 0: p * |R| seeds.
python scripts/sweep_window_large.py [n_reads] [sample_reads] [n_seeds] [gene_index] [settings "B:R,B:R,..."]"""
import os, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import bench
from megagta_amd import api, synth, hmm as hmmlib, findstart as fsm

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
sample = int(sys.argv[2]) if len(sys.argv) > 2 else 4_000_000
n_seeds = int(sys.argv[3]) if len(sys.argv) > 3 else 400_000
gi = int(sys.argv[4]) if len(sys.argv) > 4 else 0
settings = [tuple(int(x) for x in s.split(":")) for s in (sys.argv[5] if len(sys.argv) > 5 else "8192:2,8192:-1,16384:-2,32768:-4,-1:0").split(",")]
K = 45
t0 = time.time()
mg = synth.make_metagenome_device(n, 150, (("rplB", 277), ("nirK", 360)), seed=1, device="cuda:0", host_sample=sample)
ctx = api.Context(0)
rd = ctx.adopt_reads(mg.packed.data_ptr(), mg.n_words, mg.start.data_ptr(), mg.n_reads, keepalive=(mg.packed, mg.start))
ctx.keep_stream(True)
ctx.build_sdbg(rd, K - 1, collect=False)
graph = api.Graph(ctx, None, K - 1)
ctx.keep_stream(False)
ctx.release_scratch()
print(f"[{time.time() - t0:.1f} s] graph of {graph.size} edges resident", flush=True)
td = tempfile.mkdtemp(prefix="mgta_sweep_")
synth.write_gene_models(mg.genes, td)
gene = mg.genes[gi]
d = os.path.join(td, gene.name)
fw, rv = api.DeviceHmm(ctx, hmmlib.parse_hmm(os.path.join(d, "for_enone.hmm"))), api.DeviceHmm(ctx, hmmlib.parse_hmm(os.path.join(d, "rev_enone.hmm")))
words, mpos = fsm.reference_words(os.path.join(d, "ref_aligned.faa"), K // 3)
hits, _ = fsm.find_hits(ctx, rd, True, K, fsm.pack_words(words, K // 3))
ps = bench.product_seed_list(hits, mg.sample_reads, words, mpos, K)
del hits
lo = max(0, (len(ps) - n_seeds) // 2)
ps = ps[lo:lo + n_seeds]
kmers, states = [x[0] for x in ps], [x[1] - 1 for x in ps]
print(f"[{time.time() - t0:.1f} s] {gene.name}: {len(ps)} seeds of the first {sample} reads", flush=True)
for B, R in settings:
    t = time.time()
    _, offs, st = api.astar_search_packed(graph, fw, rv, kmers, states, 20, 0.5, cache_mode=B, cost_rate=R)
    dt = time.time() - t
    print(f"window {B:6d} rate {R:3d}: {dt:6.1f} s  {len(ps) / dt / 1e3:6.1f} seeds/ms  {st['n_expansions'] / 1e6:8.0f} M expansions  {st['n_expansions'] / max(1e-9, st['ms_total'] * 1e3):6.1f} M/s  "
          f"restarted {st['n_retries']} resumed {st['n_resumes']} grown {st['n_grown']} pool {st['pool_used'] / 1e9:.1f} of {st['pool_bytes'] / 1e9:.1f} GB "
          f"reserve {st['reserve_used'] / 1e9:.2f} of {st['reserve_bytes'] / 1e9:.1f} GB  largest search {st['max_search_nodes']} nodes / "
          f"{st['max_search_expansions']} expansions  contig bytes {int(offs[-1])}", flush=True)
