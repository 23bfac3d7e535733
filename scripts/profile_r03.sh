#!/bin/bash
# profiles of the round (run on the GPU box from the repo root): kernel stats of the bench command, HBM traffic of the build kernels
# (FETCH_SIZE / WRITE_SIZE in separate passes), A* counters on the 100 M-read graph (TLB, SQ).  Summaries are copied to profiles/r03/ by hand.
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/prof_r03
mkdir -p $O
export TMPDIR=/tmp
cd $R
B="python3 bench.py --no-cpu-baseline --e2e-reads 0"
echo "== kernel stats"; rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o bench -- $B --steps 5 --warmup 1 > $O/stats_line.json 2> $O/stats.err; tail -2 $O/stats.err
echo "== FETCH_SIZE"; rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/fetch -o p -- $B --seeds 0 --steps 2 --warmup 1 > /dev/null 2> $O/fetch.err; tail -1 $O/fetch.err
echo "== WRITE_SIZE"; rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/write -o p -- $B --seeds 0 --steps 2 --warmup 1 > /dev/null 2> $O/write.err; tail -1 $O/write.err
python3 scripts/pmc_traffic.py $O/fetch $O/write $O/pmc_traffic_100M_k44.json $O/traffic_latest.json > $O/pmc_traffic.txt 2>&1; head -12 $O/pmc_traffic.txt
S="$B --seeds 8000 --product-seeds 0 --steps 1 --warmup 0"
echo "== A* TLB"; rocprofv3 --pmc TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_TRANSLATION_MISS_sum --kernel-trace --output-format csv -d $O/astar_tlb -o p -- $S > $O/astar_tlb_line.json 2> $O/astar_tlb.err; tail -1 $O/astar_tlb.err
echo "== A* SQ"; rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM --kernel-trace --output-format csv -d $O/astar_sq -o p -- $S > $O/astar_sq_line.json 2> $O/astar_sq.err; tail -1 $O/astar_sq.err
echo "== A* TCC"; rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum --kernel-trace --output-format csv -d $O/astar_tcc -o p -- $S > $O/astar_tcc_line.json 2> $O/astar_tcc.err; tail -1 $O/astar_tcc.err
python3 - <<'PY'
import csv, glob, collections, json, os
O = os.environ.get("GRAFT_REPO_ROOT", os.getcwd()) + "/gpurun_out/prof_r03"
out = {}
for tag in ("astar_tlb", "astar_sq", "astar_tcc"):
    tot = collections.defaultdict(float)
    for f in glob.glob(f"{O}/{tag}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "astar_kernel" in r["Kernel_Name"]:
                tot[r["Counter_Name"]] += float(r["Counter_Value"])
    try:
        line = json.load(open(f"{O}/{tag}_line.json"))
        exp = line["search"]["expansions_per_step"] * 1.0
        # warm-up (first gene) + one step: the counters cover both
    except Exception as e:
        exp = None
    out[tag] = {"counters_sum_over_astar_dispatches": dict(tot), "expansions_of_the_timed_step": exp}
json.dump(out, open(f"{O}/astar_counters_100M.json", "w"), indent=1)
print(json.dumps(out, indent=1)[:1500])
PY
# the stats CSVs are small: keep them whole; drop the traces
find $O -name "*kernel_trace.csv" -size +1M -delete 2>/dev/null
find $O -name "*counter_collection.csv" -size +1M -delete 2>/dev/null
du -sh $O
