"""build throughput vs redundancy of the input (coverage = reads_per_genome * 150 / 20000): python scripts/bench_build_coverage.py [n_reads]"""
import sys
sys.path.insert(0, ".")
import torch  # noqa: F401
from megagta_amd import api, synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 5_000_000
ctx = api.Context(0)
for rpg in (500, 2000, 8000, 32000):
    mg = synth.make_metagenome(n, 150, (("rplB", 277),), seed=1, reads_per_genome=rpg)
    packed, start = synth.pack_reads_for_build(mg.reads)
    rd = ctx.upload_reads(packed, start)
    for it in range(2):
        s = ctx.build_sdbg(rd, 44, collect=False).stats
    print(f"coverage {rpg * 150 / 20000:6.1f}x: {s['ms_total']:7.1f} ms  {s['n_kmers'] / s['ms_total'] / 1e6:5.2f} Gk-mer/s  edges/items {s['n_edges'] / s['n_items']:.3f}  "
          f"local {s['ms_local_sort']:.1f} ms  lsd_tiles {s['n_lsd_tiles']} of {s['n_items'] // 3584}  deferred {s['n_big_segments']}", flush=True)
    rd.free()
