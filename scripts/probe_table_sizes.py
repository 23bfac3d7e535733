"""Latency of one dependent random 128-byte line (and of the same step with a 16-byte store in it) against the SIZE of the table: where the
translation reach of the device ends.  python scripts/probe_table_sizes.py [out.json]"""
import json, sys
sys.path.insert(0, ".")
from megagta_amd import api
ctx = api.Context(0)
rows = []
print("table GB | idle ns | 64 chains/CU ns | 64 chains/CU + store ns | independent peak GB/s (1024 in flight)")
for gb in (0.25, 1, 4, 16, 48, 96, 160):
    tb = int(gb * (1 << 30))
    cfg = [(1, 1, 1, 1, 4096), (8, 8, 1, 1, 8192), (8, 8, 1, 2, 8192), (16, 8, 8, 0, 2048), (8, 8, 1, 3, 8192)]
    try:
        r = ctx.probe_random_lines(tb, cfg)
    except Exception as e:
        print(gb, "failed:", e)
        break
    rows.append({"table_gb": gb, "idle_ns": r[0]["ns_per_step"], "loaded_ns": r[1]["ns_per_step"], "loaded_store_ns": r[2]["ns_per_step"], "peak_gb_per_s": r[3]["gb_per_s"], "loaded_partial_store_ns": r[4]["ns_per_step"]})
    print(f"{gb:8.2f} | {r[0]['ns_per_step']:7.1f} | {r[1]['ns_per_step']:7.1f} | {r[2]['ns_per_step']:7.1f} | {r[3]['gb_per_s']:7.1f} | partial-line store {r[4]['ns_per_step']:7.1f}", flush=True)
if len(sys.argv) > 1:
    json.dump(rows, open(sys.argv[1], "w"), indent=1)
