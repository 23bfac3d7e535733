"""`-m 2 [--need_mercy]` build timing: python scripts/bench_build_solid.py [n_reads] [k] [min_count] [mercy 0/1]"""
import json, sys, time
sys.path.insert(0, ".")
import torch  # noqa: F401
from megagta_amd import api, synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
k = int(sys.argv[2]) if len(sys.argv) > 2 else 44
m = int(sys.argv[3]) if len(sys.argv) > 3 else 2
mercy = bool(int(sys.argv[4])) if len(sys.argv) > 4 else True
mg = synth.make_metagenome(n, 150, (("rplB", 277),), seed=1)
packed, start = synth.pack_reads_for_build(mg.reads)
ctx = api.Context(0)
rd = ctx.upload_reads(packed, start)
for it in range(3):
    t = time.time()
    g = ctx.build_sdbg(rd, k, min_count=m, need_mercy=mercy, collect=False)
    dt = time.time() - t
    s = g.stats
    print(json.dumps({kk: (round(v, 2) if isinstance(v, float) else v) for kk, v in s.items()}), "wall_ms", round(dt * 1e3, 1), flush=True)
