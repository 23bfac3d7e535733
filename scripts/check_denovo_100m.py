"""denovo on the graph of the headline configuration (100 M x 150 bp, k = 44: 6.3 G edges, beyond 32-bit edge ids) - no oracle at
this size: counts, canonical contigs, and every (k+1)-mer of a sample of contigs looked up in a fresh copy of the graph.
python scripts/check_denovo_100m.py [n_reads]"""
import sys, time
sys.path.insert(0, ".")
import numpy as np
import torch  # noqa: F401
from megagta_amd import api, synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
k = 44
t0 = time.time()
mg = synth.make_metagenome(n, 150, (("rplB", 277),), seed=1)
packed, start = synth.pack_reads_for_build(mg.reads)
print(f"{n} reads generated and packed in {time.time() - t0:.0f} s", flush=True)
ctx = api.Context(0)
t0 = time.time()
stream = ctx.build_sdbg(ctx.upload_reads(packed, start), k, collect=True)
print(f"build + collect: {time.time() - t0:.1f} s, {stream.records.size} edges, {stream.stats['n_passes']} passes, device {stream.stats['ms_total']:.0f} ms", flush=True)
t0 = time.time()
g = api.Graph(ctx, stream)
print(f"graph load: {time.time() - t0:.1f} s", flush=True)
t0 = time.time()
text, st = g.denovo(150, False, k + 2)
print(f"denovo: {time.time() - t0:.1f} s wall;", {x: (round(v, 1) if isinstance(v, float) else v) for x, v in st.items()}, flush=True)
g.free()
lines = text.split("\n")
seqs = lines[1::2][: st["n_contigs"]]
comp = str.maketrans("ACGT", "TGCA")
assert len(seqs) == st["n_contigs"] and sum(map(len, seqs)) == st["total_len"]
rng = np.random.default_rng(1)
pick = rng.choice(len(seqs), size=min(2000, len(seqs)), replace=False)
assert all(seqs[i] <= seqs[i].translate(comp)[::-1] for i in pick), "a contig is not the smaller of itself and its reverse complement"
g2 = api.Graph(ctx, stream)
kmers = [seqs[i][p:p + k + 1] for i in pick for p in range(0, len(seqs[i]) - k, max(1, (len(seqs[i]) - k) // 5))]
ids = g2.index_edges(kmers)
print(f"{len(kmers)} (k+1)-mers of {len(pick)} sampled contigs looked up: {int((ids < 0).sum())} missing; max edge id {int(ids.max())}")
assert (ids >= 0).all()
# the A* leg on the same graph: every contig it returns must be a path of the graph
import os, tempfile
from megagta_amd import hmm as hmmlib
td = tempfile.mkdtemp()
synth.write_gene_models(mg.genes, td)
fw = api.DeviceHmm(ctx, hmmlib.parse_hmm(os.path.join(td, "rplB", "for_enone.hmm")))
rv = api.DeviceHmm(ctx, hmmlib.parse_hmm(os.path.join(td, "rplB", "rev_enone.hmm")))
seeds = synth.synthetic_seeds(mg.genes[0], 45, 2000, seed=4)
t0 = time.time()
res, sst = api.astar_search(g2, fw, rv, [x[0] for x in seeds], [x[1] - 1 for x in seeds], 20, 0.5)
contigs = [r.contig(x[0]).upper() for r, x in zip(res, seeds)]
# (a seed whose own (k+1)-mer no read covers is returned as it is: 0.995^45 error-free x 15x coverage leaves ~2e-4 of them uncovered)
km = [c[p:p + k + 1] for c in contigs if len(c) > k + 1 for p in range(0, len(c) - k, 7)]
ids = g2.index_edges(km)
print(f"A*: {len(seeds)} seeds, {sst['n_expansions']} expansions in {sst['ms_kernel']:.0f} ms; {len(km)} (k+1)-mers of the contigs looked up: "
      f"{int((ids < 0).sum())} missing; ids above 2^32: {int((ids >= 2**32).sum())}; mean contig {np.mean([len(c) for c in contigs]):.0f} nt")
assert (ids >= 0).all() and (ids >= 2**32).any()
print("OK")
