"""A* throughput by cache mode on real `findstart` seeds (needs the prebuilt reference at oracle/_ref for findstart only)."""
import sys, time, json, os, tempfile, subprocess
sys.path.insert(0, '.')
import numpy as np
from megagta_amd import api, synth, hmm as hmmlib
n = int(sys.argv[1]); modes = [tuple(int(y) for y in (x + ":0").split(":")[:2]) for x in sys.argv[2].split(",")]   # window[:cost_rate]
nseeds = int(sys.argv[3]) if len(sys.argv) > 3 else 0
M = 277
mg = synth.make_metagenome(n, 150, (("rplB", M),), seed=1)
td = tempfile.mkdtemp()
synth.write_lib_bin(mg.reads, os.path.join(td, "reads.lib"))
synth.write_gene_models(mg.genes, td)
t = time.time()
out = subprocess.run(["oracle/_ref/megagta", "findstart", os.path.join(td, "rplB", "ref_aligned.faa"), os.path.join(td, "reads.lib.bin"), "45", "32"],
                     stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, check=True).stdout.decode().splitlines()
seeds = [(l.split("\t")[3], int(l.split("\t")[7])) for l in out]
if nseeds: seeds = seeds[:nseeds]
print("findstart", len(seeds), "seeds", round(time.time() - t, 2), "s", flush=True)
packed, start = synth.pack_reads_for_build(mg.reads)
ctx = api.Context(0)
stream = ctx.build_sdbg(ctx.upload_reads(packed, start), 44)
g = api.Graph(ctx, stream)
fw = api.DeviceHmm(ctx, hmmlib.parse_hmm(os.path.join(td, "rplB", "for_enone.hmm")))
rv = api.DeviceHmm(ctx, hmmlib.parse_hmm(os.path.join(td, "rplB", "rev_enone.hmm")))
base = None
for mode, rate in modes:
    t = time.time()
    res, st = api.astar_search(g, fw, rv, [s[0] for s in seeds], [s[1] - 1 for s in seeds], 20, 0.5, cache_mode=mode, cost_rate=rate)
    dt = time.time() - t
    contigs = [r.contig(s[0]) for r, s in zip(res, seeds)]
    if base is None: base = contigs
    same = sum(a == b for a, b in zip(contigs, base))
    print(json.dumps({"cache_mode": mode, "cost_rate": rate, "wall_s": round(dt, 3), "ms_kernel": round(st["ms_kernel"], 1), "expansions": st["n_expansions"],
                      "Mexp_per_s": round(st["n_expansions"] / st["ms_kernel"] / 1e3, 2), "seeds_per_s": round(len(seeds) / dt, 1),
                      "retries": st["n_retries"], "same_contigs_as_first_mode": same, "distinct_contigs": len(set(contigs))}), flush=True)
