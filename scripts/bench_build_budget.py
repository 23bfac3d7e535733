"""a build under a device-memory budget (bucket-range passes), with and without the biased global-pass digits of sub-range builds:
python scripts/bench_build_budget.py [n_reads] [budget_GB] [k]      (100 M reads, 64 GB = the default of `megagta buildgraph`)"""
import os, sys
sys.path.insert(0, ".")
import torch  # noqa: F401
from megagta_amd import api, synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
gb = int(sys.argv[2]) if len(sys.argv) > 2 else 64
k = int(sys.argv[3]) if len(sys.argv) > 3 else 44
mg = synth.make_metagenome(n, 150, (("rplB", 277),), seed=1)
packed, start = synth.pack_reads_for_build(mg.reads)
print("reads ready", flush=True)
ctx = api.Context(0)
rd = ctx.upload_reads(packed, start)
ctx.set_mem_limit(gb << 30)
tot = {}
for mode in ("0", "1", "0", "1"):                       # first round warms the pool
    os.environ["MGTA_SORT_BIAS"] = mode
    s = ctx.build_sdbg(rd, k, collect=False).stats
    tot[mode] = (s["n_items"], s["n_edges"], s["n_tips"], s["n_large"])
    print(f"MGTA_SORT_BIAS={mode}: {s['n_passes']} passes, {s['n_sort_launches']} scatter launches, total {s['ms_total']:.1f} ms "
          f"(count {s['ms_count']:.0f} gen {s['ms_gen']:.0f} sort {s['ms_sort']:.0f} [scatter {s['ms_sort_scatter']:.0f} local {s['ms_local_sort']:.0f}] emit {s['ms_emit']:.0f}) "
          f"{s['n_kmers'] / s['ms_total'] / 1e6:.2f} Gk-mer/s", flush=True)
assert tot["0"] == tot["1"], tot
print("same totals with and without the bias: OK", tot["0"])
