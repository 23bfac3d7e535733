"""size-independent check at full scale: the same reads built with different memory budgets (different bucket-range splits) must give the
same totals (edges, tips, large multiplicities, items):  python scripts/check_pass_invariance.py [n_reads] [k]"""
import sys
sys.path.insert(0, ".")
import torch  # noqa: F401
from megagta_amd import api, synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
k = int(sys.argv[2]) if len(sys.argv) > 2 else 44
mg = synth.make_metagenome(n, 150, (("rplB", 277),), seed=1)
packed, start = synth.pack_reads_for_build(mg.reads)
ctx = api.Context(0)
rd = ctx.upload_reads(packed, start)
tot = []
for limit_gb in (0, 120, 70):
    ctx.set_mem_limit(limit_gb << 30)
    s = ctx.build_sdbg(rd, k, collect=False).stats
    tot.append((s["n_items"], s["n_edges"], s["n_tips"], s["n_large"]))
    print(limit_gb, "GB:", s["n_passes"], "passes", round(s["ms_total"], 1), "ms", tot[-1], flush=True)
assert len(set(tot)) == 1, tot
print("invariant across pass splits: OK")
