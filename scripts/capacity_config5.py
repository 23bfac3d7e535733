#!/usr/bin/env python3
"""BASELINE config 5's graph on ONE MI355X (what every rank of an 8-GPU run holds: the graph is replicated, the seeds are sharded): N x 150 bp
reads (default 1 G) with ten genes -> SdBG build in memory-bound bucket passes with the whole edge stream kept on the device -> the graph
packed IN PLACE into the stream's buffer (63 G edges: records + lines would be 252 GB, mgta_sdbg_load_resident) -> a cold search leg over
the ten genes with the membership check.  Reports edges, seconds and the device memory in use at the peak of every phase.
python scripts/capacity_config5.py [n_reads] [seeds_per_gene]"""
import os, sys, time, threading, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from megagta_amd import api, synth, hmm as hmmlib

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000_000
n_seeds = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
k = 44
genes = (("rplB", 277), ("nirK", 360), ("nifH", 296), ("rpoB", 240), ("amoA", 180), ("nosZ", 200), ("pmoA", 150), ("dsrA", 220), ("mcrA", 260), ("nxrB", 170))
peak = {"used": 0}
stop = False


def poll():
    while not stop:
        free, total = torch.cuda.mem_get_info()
        peak["used"] = max(peak["used"], total - free)
        time.sleep(0.05)


def phase(name, t0, extra=""):
    torch.cuda.synchronize()
    print(f"[{time.time() - T0:7.1f} s] {name}: {time.time() - t0:.1f} s, peak device memory in use so far {peak['used'] / 1e9:.1f} GB {extra}", flush=True)
    peak["used"] = 0


T0 = time.time()
threading.Thread(target=poll, daemon=True).start()
t = time.time()
mg = synth.make_metagenome_device(n, 150, genes, seed=1, device="cuda:0", host_sample=1)
torch.cuda.empty_cache()
phase(f"{n} reads generated and packed on the device", t)
ctx = api.Context(0)
rd = ctx.adopt_reads(mg.packed.data_ptr(), mg.n_words, mg.start.data_ptr(), mg.n_reads, keepalive=(mg.packed, mg.start))
ctx.keep_stream(True)
t = time.time()
st = ctx.build_sdbg(rd, k, collect=False).stats
phase("SdBG build (whole stream kept on the device)", t, f"| {st['n_passes']} passes, {st['n_items']} sort items, {st['n_edges']} edges, device {st['ms_total']:.0f} ms = "
      f"{st['n_kmers'] / st['ms_total'] / 1e6:.2f} Gk-mer/s")
genes_meta = mg.genes
rd.free(); rd._keep = None
mg.packed = mg.start = None
torch.cuda.empty_cache()
t = time.time()
graph = api.Graph(ctx, None, k)                       # (short of memory for a second copy: the lines go INTO the stream's buffer)
ctx.keep_stream(False)
ctx.release_scratch()
phase(f"graph of {graph.size} edges resident", t)
td = tempfile.mkdtemp()
synth.write_gene_models(genes_meta, td)
t = time.time()
tot_e = 0
found = sampled = above = 0
import bench
for gi, gene in enumerate(genes_meta if n_seeds > 0 else []):
    d = os.path.join(td, gene.name)
    fw, rv = api.DeviceHmm(ctx, hmmlib.parse_hmm(os.path.join(d, "for_enone.hmm"))), api.DeviceHmm(ctx, hmmlib.parse_hmm(os.path.join(d, "rev_enone.hmm")))
    seeds = synth.synthetic_seeds(gene, 45, n_seeds, seed=4 + gi)
    tg = time.time()
    cont, offs, s = api.astar_search_packed(graph, fw, rv, [x[0] for x in seeds], [x[1] - 1 for x in seeds], 20, 0.5)
    tot_e += s["n_expansions"]
    m = bench.contig_membership(graph, {0: (cont, offs)}, k, n_sample=10000)
    sampled += m["sampled"]; found += m.get("found", 0); above += m.get("ids_above_2^32", 0)
    print(f"[{time.time() - T0:7.1f} s]   {gene.name}: {s['n_expansions']} expansions in {time.time() - tg:.1f} s, membership {m.get('found', 0)} of {m['sampled']}", flush=True)
    fw.free(); rv.free()
phase(f"cold search, {len(genes_meta)} genes x {n_seeds} seeds: {tot_e} expansions", t, f"| membership {found} of {sampled} (k+1)-mers found, {above} ids above 2^32")
stop = True
