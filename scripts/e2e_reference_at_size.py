"""The reference beside ours ABOVE the bench's same-sample size: `megagta.py -k 30,36,45` (rplB + nirK) on N synthetic reads, ours and the
reference binary (oracle/_ref/megagta behind the same driver, 32 threads) on the SAME files, contigs compared as multisets.
python scripts/e2e_reference_at_size.py [n_reads] [out.json] [hard stop seconds]  (a gpurun call ends at 1200 s: 10 M reads fit, 20 M do not)"""
import json, os, sys, time
sys.path.insert(0, ".")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
out = sys.argv[2] if len(sys.argv) > 2 else None
stop = float(sys.argv[3]) if len(sys.argv) > 3 else 1100.0
os.environ.setdefault("MEGAGTA_E2E_REF_THREADS", "32")
os.environ.setdefault("MEGAGTA_E2E_SKIP_UNORDERED", "1")
os.environ["MEGAGTA_E2E_KEEP_CONTENT"] = "1"
if out:
    os.makedirs(os.path.dirname(out) or ".", exist_ok=True)
    os.environ.setdefault("MEGAGTA_E2E_LOG_DIR", os.path.dirname(out) or ".")
t0 = time.time()
import threading


def heartbeat():                                     # (the reference's run is silent for minutes: gpurun takes 7 silent minutes for a hang)
    while True:
        time.sleep(60)
        print(f"[e2e_reference_at_size {time.time() - t0:6.0f} s] still running", file=sys.stderr, flush=True)


threading.Thread(target=heartbeat, daemon=True).start()
import torch  # noqa: F401  (one HIP runtime for torch and the library)
import bench
res = bench.e2e_leg((("rplB", 277), ("nirK", 360)), n, 1, "cuda:0", hard_stop=t0 + stop)
res["host"] = bench.host_cores()
res["wall_s"] = time.time() - t0
try:
    import subprocess
    res["commit"] = subprocess.run(["git", "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip() or None
except Exception:
    pass
res["date"] = time.strftime("%Y-%m-%d")
print(json.dumps(res))
if out:
    json.dump(res, open(out, "w"), indent=1)
