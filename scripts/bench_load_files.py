#!/usr/bin/env python3
"""How long a rank takes to load a graph from its files ("a rank loads a 630 M-edge graph in seconds"):
reads generated on the device -> reads.lib.bin -> `megagta buildgraph` (one process, and as N ranks into N files) -> mgta_sdbg_load_files, timed.
python scripts/bench_load_files.py [n_reads=10000000] [ranks=4]"""
import os, subprocess, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from megagta_amd import api, synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
ranks = int(sys.argv[2]) if len(sys.argv) > 2 else 4
BIN = os.path.join(ROOT, "megagta_amd", "bin", "megagta")
d = tempfile.mkdtemp(prefix="mgta_load_")
mg = synth.make_metagenome_device(n, 150, (("rplB", 277),), seed=1, device="cuda:0", host_sample=n)
synth.write_lib_bin(mg.sample_reads, d + "/reads.lib")
del mg
torch.cuda.empty_cache()
common = ["-k", "44", "-m", "1", "--host_mem", "64000000000", "--mem_flag", "1", "--gpu_mem", "0", "--num_cpu_threads", "8", "--num_output_threads", "1",
          "--read_lib_file", d + "/reads.lib"]
t = time.time()
subprocess.run([BIN, "buildgraph", "--output_prefix", d + "/one"] + common, check=True, capture_output=True)
print(f"buildgraph, one process: {time.time() - t:.1f} s, {os.path.getsize(d + '/one.sdbg.0') / 1e9:.2f} GB", flush=True)
t = time.time()
ps = [subprocess.Popen([BIN, "buildgraph", "--output_prefix", d + "/many"] + common, stderr=subprocess.DEVNULL,
                       env={**os.environ, "MEGAGTA_RANK": str(r), "MEGAGTA_WORLD": str(ranks), "MEGAGTA_DEVICE": "0"}) for r in range(ranks)]
assert all(p.wait() == 0 for p in ps)
subprocess.run([BIN, "sdbgmerge", d + "/many", str(ranks)], check=True)
print(f"buildgraph as {ranks} ranks on one GPU + sdbgmerge: {time.time() - t:.1f} s", flush=True)
ctx = api.Context(0)
for name in ("one", "many", "one", "many"):
    t = time.time()
    g = api.Graph.from_files(ctx, d + "/" + name)
    dt = time.time() - t
    print(f"mgta_sdbg_load_files({name}): {g.size} edges in {dt:.2f} s = {g.size * 2 / dt / 1e9:.2f} GB/s of records", flush=True)
    g.free()
import shutil
shutil.rmtree(d, ignore_errors=True)
