"""Random 128-byte line ceiling of the device (mgta_probe_random_lines): independent lines swept over lines in flight per CU, and the
loaded latency of one dependent line.  python scripts/probe_random_lines.py [table GB] [out.json]"""
import json, sys
sys.path.insert(0, ".")
from megagta_amd import api

gb = float(sys.argv[1]) if len(sys.argv) > 1 else 16.0
out = sys.argv[2] if len(sys.argv) > 2 else None
ctx = api.Context(0)
tb = int(gb * (1 << 30))


def run(cfgs, target_ms=40.0):
    """each configuration twice: a short run sizes the steps of the timed one"""
    first = ctx.probe_random_lines(tb, [(w, g, u, d, 256) for (w, g, u, d) in cfgs])
    sized = []
    for c, r in zip(cfgs, first):
        steps = max(256, int(256 * target_ms / max(r["ms"], 1e-3)))
        sized.append((*c, min(steps, 1 << 22)))
    return ctx.probe_random_lines(tb, sized)


indep = []
for lif in (4, 8, 16, 32, 64, 128, 256, 512, 1024, 2048):
    # lines in flight per CU = waves * groups * unroll: fill groups first (the kernels' wavefronts carry 8 searches), then waves, then unroll
    g = min(8, lif); w = min(32 if lif > 1024 else 16, max(1, lif // g)); u = max(1, lif // (g * w))
    indep.append((w, g, u, 0))
indep += [(8, 8, 1, 0), (32, 8, 1, 0), (32, 8, 2, 0), (32, 8, 4, 0), (8, 8, 8, 0), (16, 8, 8, 0)]
dep = [(1, 1, 1, 1), (1, 8, 1, 1), (8, 8, 1, 1), (16, 8, 1, 1), (32, 8, 1, 1), (8, 8, 4, 1), (16, 8, 4, 1), (32, 8, 4, 1),
       (1, 1, 1, 2), (1, 8, 1, 2), (8, 8, 1, 2), (16, 8, 1, 2)]           # ... and the chase with a 16-byte store to a random line in every step
res_i = run(indep)
res_d = run(dep)
print(f"table {gb:.0f} GB of 128-byte lines; groups of 8 lanes read one line each")
print("independent lines:  waves/CU groups unroll | lines in flight/CU |   GB/s   | G lines/s")
for r in res_i:
    print(f"   {r['waves_per_cu']:3d} {r['groups']:2d} {r['unroll']:2d} | {r['lines_in_flight_per_cu']:5d} | {r['gb_per_s']:8.1f} | {r['gb_per_s'] / 128:6.2f}")
print("dependent chains:   waves/CU groups chains | chains/CU |  ns per dependent line | GB/s")
for r in res_d:
    print(f"   {r['waves_per_cu']:3d} {r['groups']:2d} {r['unroll']:2d} | {r['lines_in_flight_per_cu']:5d} | {r['ns_per_step']:8.1f} | {r['gb_per_s']:8.1f}" + ("   (+ a store per step)" if r["dependent"] == 2 else ""))
best = max(res_i, key=lambda r: r["gb_per_s"])
doc = {"table_bytes": tb, "independent": res_i, "dependent": res_d, "peak_gb_per_s": best["gb_per_s"],
       "peak_at_lines_in_flight_per_cu": best["lines_in_flight_per_cu"], "idle_dependent_ns": res_d[0]["ns_per_step"]}
if out:
    json.dump(doc, open(out, "w"), indent=1)
print(json.dumps({k: v for k, v in doc.items() if k not in ("independent", "dependent")}))
