#!/usr/bin/env python3
"""device window-1 search vs the oracle's sequential search on the 1 M-read parity input: first differing seed and its details"""
import os, sys, time, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from megagta_amd import api, synth, hmm as hmmlib
from oracle import oracle as O
O.build()
BIN = os.path.join(ROOT, "megagta_amd", "bin", "megagta")
d = "/tmp/p1m"
os.makedirs(d, exist_ok=True)
gname = sys.argv[1] if len(sys.argv) > 1 else "rplB"
n_take = int(sys.argv[2]) if len(sys.argv) > 2 else 6000
mg = synth.make_metagenome(1_000_000, 150, (("rplB", 277), ("nirK", 360)), seed=77)
synth.write_lib_bin(mg.reads, d + "/reads.lib")
gl = synth.write_gene_models(mg.genes, d + "/models")
genes = {l.split()[0]: l.split() for l in open(gl)}
a = genes[gname]
lines = subprocess.run([BIN, "findstart", a[3], d + "/reads.lib.bin", "45", "4"], check=True, capture_output=True).stdout.decode().splitlines()
step = len(lines) // n_take
lines = lines[::step][:n_take]
seeds = [(l.split("\t")[3], int(l.split("\t")[7])) for l in lines]
packed, start = synth.pack_reads_for_build(mg.reads)
ctx = api.Context(0)
stream = ctx.build_sdbg(ctx.upload_reads(packed, start), 44)
g = api.Graph(ctx, stream)
fw, rv = api.DeviceHmm(ctx, hmmlib.parse_hmm(a[1])), api.DeviceHmm(ctx, hmmlib.parse_hmm(a[2]))
km, ss = [s[0] for s in seeds], [s[1] - 1 for s in seeds]
res1, st1 = api.astar_search(g, fw, rv, km, ss, 20, 0.5, cache_mode=1)
print("device window 1:", st1["n_expansions"], "expansions", flush=True)
og = O.Graph(O.Stream.build(packed, start, 44, threads=16))
S = O.Searcher(og, O.Hmm(a[1]), O.Hmm(a[2]), 20, 0.5)
S.clear_cache(); S.set_window(1); S.set_cost_rate(0)
nd = 0
t = time.time()
for i, (k_, p_) in enumerate(seeds):
    contig, R, L = S.search(k_, p_ - 1, cold=False)
    r = res1[i]
    same = r.contig(k_) == contig
    cnt = (r.right_side["n_closed"], r.right_side["n_expanded"], r.left_side["n_closed"], r.left_side["n_expanded"]) == (R.n_closed, R.n_expanded, L.n_closed, L.n_expanded)
    if not same or not cnt:
        nd += 1
        if nd <= 8:
            print(f"seed {i} {k_} pos {p_}: contig equal {same}; device R closed/expanded {r.right_side['n_closed']}/{r.right_side['n_expanded']} L {r.left_side['n_closed']}/{r.left_side['n_expanded']}; "
                  f"oracle R {R.n_closed}/{R.n_expanded} L {L.n_closed}/{L.n_expanded}; scores dev {r.right_side['real_score']:.6f} {r.left_side['real_score']:.6f} oracle {R.real_score:.6f} {L.real_score:.6f}", flush=True)
            if not same:
                dc, oc = r.contig(k_), contig
                print("   len", len(dc), len(oc), "first diff at", next((j for j, (x, y) in enumerate(zip(dc, oc)) if x != y), -1), flush=True)
print(f"{gname}: {len(seeds)} seeds, {nd} differ from the oracle's sequential run ({time.time() - t:.0f} s of oracle)", flush=True)
# the same seeds cold on the device vs cold oracle for the first differing ones would need minutes of CPU: skipped
