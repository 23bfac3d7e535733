#!/bin/bash
# A/B runs of the bench's build leg (100 M reads, k = 44) under rocprofv3 kernel stats: one run per argument "tag:NAME=VALUE,NAME=VALUE"
# (e.g. "side0:MGTA_SORT_SIDE=0" "side1:" "ipt8:MEGAGTA_HIP_LIB=$PWD/megagta_amd/libmegagta_hip_ipt8.so").  Run on the GPU box from the
# repo root; the bench lines and the kernel stats land in gpurun_out/ab_build.
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/ab_build
mkdir -p $O
export TMPDIR=/tmp
cd $R
B="python3 bench.py --no-cpu-baseline --e2e-reads 0 --seeds 0 --steps 5 --warmup 1"
for v in "$@"; do
  tag=${v%%:*}; envs=${v#*:}
  ( IFS=, read -ra kvs <<< "$envs"; for kv in "${kvs[@]}"; do [ -n "$kv" ] && export "$kv"; done
    echo "== $tag ($envs)"
    rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_$tag -o bench -- $B > $O/line_$tag.json 2> $O/stats_$tag.err || { tail -5 $O/stats_$tag.err; exit 1; }
    tail -1 $O/line_$tag.json | cut -c1-330 ) || exit 1
  find $O -name "*kernel_trace.csv" -size +1M -delete 2>/dev/null
  find $O -name "*.db" -delete 2>/dev/null
  python3 - $O/stats_$tag/bench_kernel_stats.csv <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if "mgta::" in r["Name"] and float(r["TotalDurationNs"]) > 3e7:
        print(f'   {r["Name"][:66]:66s} {r["Calls"]:>4s} x {float(r["AverageNs"])/1e6:8.3f} ms = {float(r["TotalDurationNs"])/1e6/6:7.1f} ms/step')
PY
done
