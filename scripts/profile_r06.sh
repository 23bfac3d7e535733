#!/bin/bash
# profiles of round 6 (run on the GPU box from the repo root; `bash scripts/profile_r06.sh [build|astar|all]`): kernel stats of the bench
# command, HBM traffic of the build kernels (FETCH_SIZE / WRITE_SIZE in separate passes), A* counters on the 100 M-read graph (SQ, TCC).
# Counter passes carry --kernel-trace only (never --stats / sys / hip traces with --pmc).  Summaries land in gpurun_out/prof_r06 and are
# copied to profiles/r06/ (+ profiles/traffic_latest.json, profiles/astar_counters_latest.json, which bench.py quotes only while their
# `source_signature` matches the kernel sources).
what=${1:-all}
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/prof_r06
mkdir -p $O
export TMPDIR=/tmp
cd $R
B="python3 bench.py --no-cpu-baseline --e2e-reads 0"
if [ "$what" = "build" ] || [ "$what" = "all" ]; then
  echo "== kernel stats (build only)"; rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o bench -- $B --seeds 0 --steps 5 --warmup 1 > $O/stats_line.json 2> $O/stats.err; tail -2 $O/stats.err
  echo "== FETCH_SIZE"; rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/fetch -o p -- $B --seeds 0 --steps 2 --warmup 1 > /dev/null 2> $O/fetch.err; tail -1 $O/fetch.err
  echo "== WRITE_SIZE"; rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/write -o p -- $B --seeds 0 --steps 2 --warmup 1 > /dev/null 2> $O/write.err; tail -1 $O/write.err
  python3 scripts/pmc_traffic.py $O/fetch $O/write $O/pmc_traffic_100M_k44.json $O/traffic_latest.json > $O/pmc_traffic.txt 2>&1; head -12 $O/pmc_traffic.txt
  python3 - <<'PY'
import json, os, subprocess, sys, time
R = os.environ.get("GRAFT_REPO_ROOT", os.getcwd()); sys.path.insert(0, R)
import bench
p = R + "/gpurun_out/prof_r06/traffic_latest.json"
j = json.load(open(p))
j.update({"reads": 100000000, "graph_k": 44, "source_signature": bench.source_signature(*bench.BUILD_SOURCES), "collected": time.strftime("%Y-%m-%d %H:%M UTC", time.gmtime()),
          "commit": os.environ.get("MGTA_COMMIT", "?"), "_source": "profiles/r06/pmc_traffic_100M_k44.json (2 * FETCH_SIZE + WRITE_SIZE)"})
json.dump(j, open(p, "w"), indent=1)
PY
fi
if [ "$what" = "astar" ] || [ "$what" = "all" ]; then
  S="$B --seeds 60000 --product-seeds 0 --steps 1 --warmup 0"
  echo "== A* kernel stats"; rocprofv3 --kernel-trace --stats --output-format csv -d $O/astar_stats -o p -- $S > $O/astar_stats_line.json 2> $O/astar_stats.err; tail -1 $O/astar_stats.err
  echo "== A* SQ"; rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM --kernel-trace --output-format csv -d $O/astar_sq -o p -- $S > $O/astar_sq_line.json 2> $O/astar_sq.err; tail -1 $O/astar_sq.err
  echo "== A* TCC"; rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum --kernel-trace --output-format csv -d $O/astar_tcc -o p -- $S > $O/astar_tcc_line.json 2> $O/astar_tcc.err; tail -1 $O/astar_tcc.err
  python3 - <<'PY'
import csv, glob, collections, json, os, sys, time
R = os.environ.get("GRAFT_REPO_ROOT", os.getcwd()); sys.path.insert(0, R)
import bench
O = R + "/gpurun_out/prof_r06"
out = {}
for tag in ("astar_sq", "astar_tcc"):
    tot = collections.defaultdict(float)
    for f in glob.glob(f"{O}/{tag}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "astar_kernel" in r["Kernel_Name"]:
                tot[r["Counter_Name"]] += float(r["Counter_Value"])
    exp = None
    try:
        line = json.load(open(f"{O}/{tag}_line.json"))
        # the counters cover the warm-up (first gene) and the timed step: both run the same seeds, so expansions = step + first gene's share
        exp = line["search"]["expansions_per_step"] + line["search"].get("warmup_expansions", 0)
    except Exception as e:
        print("no bench line for", tag, e)
    out[tag] = {"counters_sum_over_astar_dispatches": dict(tot), "expansions_of_all_astar_dispatches": exp}
json.dump(out, open(f"{O}/astar_counters_100M.json", "w"), indent=1)
t = out["astar_tcc"]
if t["expansions_of_all_astar_dispatches"]:
    latest = {"reads": 100000000, "graph_k": 44, "lanes_per_search": 8, "TCC_MISS_sum": t["counters_sum_over_astar_dispatches"].get("TCC_MISS_sum"),
              "TCC_HIT_sum": t["counters_sum_over_astar_dispatches"].get("TCC_HIT_sum"), "expansions": t["expansions_of_all_astar_dispatches"],
              "command": "rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum --kernel-trace -- python3 bench.py --no-cpu-baseline --e2e-reads 0 --seeds 60000 --product-seeds 0 --steps 1 --warmup 0",
              "source_signature": bench.source_signature(*bench.ASTAR_SOURCES), "collected": time.strftime("%Y-%m-%d %H:%M UTC", time.gmtime()), "commit": os.environ.get("MGTA_COMMIT", "?")}
    json.dump(latest, open(f"{O}/astar_counters_latest.json", "w"), indent=1)
print(json.dumps(out, indent=1)[:1800])
PY
fi
find $O -name "*kernel_trace.csv" -size +1M -delete 2>/dev/null
find $O -name "*counter_collection.csv" -size +1M -delete 2>/dev/null
find $O -name "*.db" -delete 2>/dev/null
du -sh $O
