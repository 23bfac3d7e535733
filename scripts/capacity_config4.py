#!/usr/bin/env python3
"""BASELINE config 4's graph on ONE MI355X: N x 150 bp reads (default 500 M) with five genes -> SdBG build in memory-bound bucket passes with
the whole edge stream kept on the device -> graph resident (mgta_sdbg_load_resident) -> a cold search leg over all five genes with the
membership check -> denovo (tips, bubbles, unitigs) on the same graph.  Reports edges, seconds and the device memory in use at the peak
of every phase (polled through hipMemGetInfo from a second thread).  python scripts/capacity_config4.py [n_reads] [seeds_per_gene]"""
import os, sys, time, threading, tempfile, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from megagta_amd import api, synth, hmm as hmmlib

n = int(sys.argv[1]) if len(sys.argv) > 1 else 500_000_000
n_seeds = int(sys.argv[2]) if len(sys.argv) > 2 else 4000      # 0: skip the search leg
os.environ.setdefault("MGTA_DENOVO_VERBOSE", "1")               # one line per phase / every 16 bubble rounds on stderr: minutes of work otherwise look like a hang
k = 44
genes = (("rplB", 277), ("nirK", 360), ("nifH", 296), ("rpoB", 240), ("amoA", 180))
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
t = time.time()
graph = api.Graph(ctx, None, k)
ctx.keep_stream(False)
ctx.release_scratch()
phase(f"graph of {graph.size} edges resident", t)
td = tempfile.mkdtemp()
synth.write_gene_models(mg.genes, td)
t = time.time()
tot_e = 0
found = sampled = above = 0
for gi, gene in enumerate(mg.genes if n_seeds > 0 else []):
    d = os.path.join(td, gene.name)
    fw, rv = api.DeviceHmm(ctx, hmmlib.parse_hmm(os.path.join(d, "for_enone.hmm"))), api.DeviceHmm(ctx, hmmlib.parse_hmm(os.path.join(d, "rev_enone.hmm")))
    seeds = synth.synthetic_seeds(gene, 45, n_seeds, seed=4 + gi)
    cont, offs, s = api.astar_search_packed(graph, fw, rv, [x[0] for x in seeds], [x[1] - 1 for x in seeds], 20, 0.5)
    tot_e += s["n_expansions"]
    import bench
    m = bench.contig_membership(graph, {0: (cont, offs)}, k, n_sample=20000)
    sampled += m["sampled"]; found += m.get("found", 0); above += m.get("ids_above_2^32", 0)
    fw.free(); rv.free()
phase(f"cold search, 5 genes x {n_seeds} seeds: {tot_e} expansions", t, f"| membership {found} of {sampled} (k+1)-mers found, {above} ids above 2^32")
ctx.release_scratch()
t = time.time()
# contigs of >= 300 characters only: every phase runs on every path, but the text that comes back to the host stays small (all contigs of
# a 31.6 G-edge graph are ~30 GB of FASTA, held three times on the way to Python)
import ctypes as C
from megagta_amd._lib import DenovoStats, check
txt, ln, dst = C.c_void_p(), C.c_uint64(), DenovoStats()
check(ctx._L.mgta_denovo(graph.h, 150, 0, 300, C.byref(txt), C.byref(ln), C.byref(dst)), "mgta_denovo")
ctx._L.mgta_host_free(txt)
dst = dst.as_dict()
phase("denovo", t, "| " + json.dumps({x: (round(v, 1) if isinstance(v, float) else v) for x, v in dst.items()}))
print(f"contig text (contigs >= 300) {ln.value / 1e9:.2f} GB", flush=True)
stop = True
