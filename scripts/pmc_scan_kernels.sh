#!/bin/bash
# SQ counters of the build's kernels on the bench's build leg (100 M reads): instructions per wave and busy shares, per kernel name.
# Counter passes carry --kernel-trace only.  Output: gpurun_out/pmc_scan/summary.txt
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/pmc_scan
mkdir -p $O
export TMPDIR=/tmp
cd $R
B="python3 bench.py --no-cpu-baseline --e2e-reads 0 --seeds 0 --steps 1 --warmup 0"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM --kernel-trace --output-format csv -d $O/a -o p -- $B > /dev/null 2> $O/a.err || { tail -3 $O/a.err; exit 1; }
rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAVES SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $O/b -o p -- $B > /dev/null 2> $O/b.err || { tail -3 $O/b.err; exit 1; }
python3 - <<'PY' > $O/summary.txt
import csv, glob, collections, os
O = os.environ.get("GRAFT_REPO_ROOT", os.getcwd()) + "/gpurun_out/pmc_scan"
tot = collections.defaultdict(lambda: collections.defaultdict(float)); calls = collections.Counter()
for tag in "ab":
    for f in glob.glob(f"{O}/{tag}/**/*counter_collection.csv", recursive=True):
        seen = set()
        for r in csv.DictReader(open(f)):
            n = r["Kernel_Name"].split("(")[0][-60:]
            if "mgta::" not in r["Kernel_Name"]: continue
            tot[n][r["Counter_Name"]] += float(r["Counter_Value"])
            if tag == "a" and r["Counter_Name"] == "SQ_INSTS_VALU": calls[n] += 1
for n, c in sorted(tot.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", 0)):
    w = max(1.0, c.get("SQ_WAVES", 0))
    print(f"{n:62s} launches {calls[n]:3d} waves {w:.3g} | per wave: VALU {c.get('SQ_INSTS_VALU',0)/w:8.0f} SALU {c.get('SQ_INSTS_SALU',0)/w:8.0f} VMEM {c.get('SQ_INSTS_VMEM',0)/w:6.0f} LDS {c.get('SQ_INSTS_LDS',0)/w:6.0f}"
          f" | wave cycles {c.get('SQ_WAVE_CYCLES',0):.3g} wait_any {c.get('SQ_WAIT_ANY',0)/max(1,c.get('SQ_WAVE_CYCLES',0)):.2f} active_any {c.get('SQ_ACTIVE_INST_ANY',0)/max(1,c.get('SQ_WAVE_CYCLES',0)):.2f}"
          f" | busy cycles {c.get('SQ_BUSY_CYCLES',0):.3g} active_valu {c.get('SQ_ACTIVE_INST_VALU',0):.3g}")
PY
find $O -name "*counter_collection.csv" -size +1M -delete 2>/dev/null; find $O -name "*kernel_trace.csv" -size +1M -delete 2>/dev/null; find $O -name "*.db" -delete 2>/dev/null
cat $O/summary.txt
