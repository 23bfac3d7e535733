"""timing of the sort routes (diagnostic): python scripts/bench_sort_modes.py [n_reads] [k] [modes...]"""
import sys
sys.path.insert(0, ".")
import torch  # noqa: F401  (before the HIP library)
from megagta_amd import api, synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10000000
k = int(sys.argv[2]) if len(sys.argv) > 2 else 44
modes = [int(x) for x in sys.argv[3:]] or [0, 2]
mg = synth.make_metagenome(n, 150, (("rplB", 277),), seed=1)
packed, start = synth.pack_reads_for_build(mg.reads)
ctx = api.Context(0)
rd = ctx.upload_reads(packed, start)
for mode in modes:
    ctx.set_full_lsd(mode)
    for it in range(2):
        g = ctx.build_sdbg(rd, k, collect=False)
    s = g.stats
    print(mode, "local", round(s["ms_local_sort"], 2), "scatter", round(s["ms_sort_scatter"], 2), "sort", round(s["ms_sort"], 2), "total", round(s["ms_total"], 2),
          "lsd_tiles", s["n_lsd_tiles"], "deferred", s["n_big_segments"], "Gkmer/s", round(s["n_kmers"] / s["ms_total"] / 1e6, 3), flush=True)
