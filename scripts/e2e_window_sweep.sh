for cfg in "2048 2" "4096 2" "4096 4" "8192 4" "2048 4" "1024 4" "8192 2"; do
  set -- $cfg
  MEGAGTA_CACHE_WINDOW=$1 MEGAGTA_CACHE_COST_RATE=$2 MEGAGTA_E2E_SKIP_UNORDERED=1 MEGAGTA_E2E_LOG_DIR=gpurun_out timeout -k 10 200 python bench.py --reads 1000000 --seeds 0 --steps 1 --no-cpu-baseline --e2e-ref-reads 0 > gpurun_out/sw.log 2>gpurun_out/sw.err
  echo "window $1 rate $2: $(grep -a 'e2e ours' gpurun_out/sw.err | sed 's/.*e2e ours: //') | $(grep -a 'Done ' gpurun_out/e2e_ours.log | sed 's/.*Done \([a-zA-Z]*\): time \([0-9.]*\) (\([0-9]*\) expansions.*/\1 \2 s \3/' | tr '\n' ' ')" >> gpurun_out/e2e_window_sweep.log
  tail -1 gpurun_out/e2e_window_sweep.log
done
