import sys, time, json
sys.path.insert(0, '.')
import numpy as np
from megagta_amd import api, synth
n = int(sys.argv[1]); k = int(sys.argv[2])
t = time.time()
mg = synth.make_metagenome(n, 150, (("rplB", 277),), seed=1)
packed, start = synth.pack_reads_for_build(mg.reads)
print("gen", time.time() - t, flush=True)
ctx = api.Context(0)
rd = ctx.upload_reads(packed, start)
if len(sys.argv) > 3:
    ctx.set_full_lsd(int(sys.argv[3]))
for it in range(3):
    g = ctx.build_sdbg(rd, k, collect=False)
    s = g.stats
    print(json.dumps({kk: (round(v, 3) if isinstance(v, float) else v) for kk, v in s.items()}), flush=True)
    print("Gkmer/s", s["n_kmers"] / s["ms_total"] / 1e6, "scatter GB/s", s["n_items"] * s["words_per_key"] * 4 * 2 * s["n_sort_launches"] / s["n_passes"] / s["ms_sort_scatter"] / 1e6, flush=True)
