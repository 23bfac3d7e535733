# window / cost-rate sweep of `megagta search` on bench.py's e2e workload: bash scripts/e2e_window_sweep.sh [reads] ["B R [lanes]" ...]
reads=${1:-2000000}; shift
[ $# -eq 0 ] && set -- "2048 2" "4096 2" "4096 4" "8192 4" "2048 4" "1024 4" "8192 2"
for cfg in "$@"; do
  set -- $cfg
  if [ "$1" = default ]; then unset MEGAGTA_CACHE_WINDOW MEGAGTA_CACHE_COST_RATE; else export MEGAGTA_CACHE_WINDOW=$1 MEGAGTA_CACHE_COST_RATE=$2; fi
  if [ -n "$3" ]; then export MGTA_ASTAR_GROUP=$3; else unset MGTA_ASTAR_GROUP; fi
  MEGAGTA_E2E_SKIP_UNORDERED=1 MEGAGTA_E2E_LOG_DIR=gpurun_out timeout -k 10 300 python bench.py --reads 1000000 --seeds 0 --steps 1 --no-cpu-baseline --e2e-reads $reads --e2e-ref-reads 0 > gpurun_out/sw.log 2>gpurun_out/sw.err
  echo "$reads reads, window $1 rate $2 lanes ${3:-auto}: $(grep -a 'e2e ours' gpurun_out/sw.err | sed 's/.*e2e ours: //') | $(grep -a 'Done ' gpurun_out/e2e_ours.log | sed 's/.*Done \([a-zA-Z]*\): time \([0-9.]*\) (\([0-9]*\) expansions.*/\1 \2 s \3/' | tr '\n' ' ')" >> gpurun_out/e2e_window_sweep_$reads.log
  tail -1 gpurun_out/e2e_window_sweep_$reads.log
done
