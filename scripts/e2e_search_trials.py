#!/usr/bin/env python3
"""The search step of a multi-k run at scale, tried under several settings on ONE set of artefacts: the pipeline up to the seeds is run
once (one process per step: reads.fa -> buildlib -> buildgraph 29 -> denovo -> buildgraph 35 -> denovo -> buildgraph 44 -> findstart),
then `megagta search` runs on the first `n_seeds` seeds of one gene once per setting, each under a time limit, with the batch monitor on.
The graph of a multi-k run carries the previous k's contigs as assist sequences and is far more branched around the genes than the graph
of the reads alone (bench.py's): the search's behaviour at 50-100 M reads can only be studied on it.

python scripts/e2e_search_trials.py <n_reads> <gene> <n_seeds> <seconds per trial> "NAME=VAL,NAME=VAL;NAME=VAL;..." [log dir]"""
import os, shutil, subprocess, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from megagta_amd import synth

n = int(sys.argv[1])
gene = sys.argv[2]
n_seeds = int(sys.argv[3])
limit = int(sys.argv[4])
trials = [dict(kv.split("=") for kv in t.split(",") if kv) for t in sys.argv[5].split(";")]
logdir = sys.argv[6] if len(sys.argv) > 6 else os.path.join(ROOT, "gpurun_out")
os.makedirs(logdir, exist_ok=True)
BIN = os.path.join(ROOT, "megagta_amd", "bin", "megagta")
d = tempfile.mkdtemp(prefix="mgta_trials_")
t00 = time.time()


def step(cmd, stdout=None, env=None):
    t = time.time()
    r = subprocess.run(cmd, stdout=stdout if stdout else subprocess.DEVNULL, stderr=subprocess.PIPE, text=True, env=env)
    if r.returncode != 0:
        sys.exit(f"{cmd[1]} failed: {r.stderr[-1500:]}")
    print(f"[{time.time() - t00:6.1f} s] {cmd[1]} {time.time() - t:.1f} s", flush=True)
    return r


try:
    fa = open(d + "/reads.fa", "wb")
    L, width = 150, 9
    lut = np.frombuffer(b"ACGT", dtype=np.uint8)

    def sink(first, codes):
        m = codes.shape[0]
        rec = np.empty((m, 2 + width + 1 + L + 1), dtype=np.uint8)
        rec[:, 0], rec[:, 1] = ord(">"), ord("r")
        ids = np.arange(first, first + m, dtype=np.int64)
        for dg in range(width):
            rec[:, 2 + width - 1 - dg] = (ids % 10 + ord("0")).astype(np.uint8)
            ids //= 10
        rec[:, 2 + width] = ord("\n")
        rec[:, 3 + width:3 + width + L] = lut[codes]
        rec[:, -1] = ord("\n")
        rec.tofile(fa)

    mg = synth.make_metagenome_device(n, L, (("rplB", 277), ("nirK", 360)), seed=1000 + n % 997, device="cuda:0", host_sample=0, on_chunk=sink)
    fa.close()
    gl = synth.write_gene_models(mg.genes, d + "/models")
    del mg
    import torch
    torch.cuda.empty_cache()
    print(f"[{time.time() - t00:6.1f} s] reads.fa written", flush=True)
    open(d + "/reads.lib", "w").write(f"reads.fa\nse {d}/reads.fa\n")
    step([BIN, "buildlib", d + "/reads.lib", d + "/reads.lib"])
    common = ["-m", "1", "--host_mem", "100000000000", "--mem_flag", "1", "--gpu_mem", str(170 << 30), "--num_cpu_threads", "16", "--num_output_threads", "5", "--read_lib_file", d + "/reads.lib"]
    prev = None
    for k, nxt in ((29, 35), (35, 44), (44, None)):
        cmd = [BIN, "buildgraph", "-k", str(k), "--output_prefix", f"{d}/{k}"] + common
        if prev:
            cmd += ["--assist_seq", f"{d}/{prev}.contigs.fa"]
        step(cmd)
        if nxt:
            step([BIN, "denovo", "-s", f"{d}/{k}", "-o", f"{d}/{k}", "-t", "16", "--min_standalone", "400", "--max_tip_len", "150", "--min_contig", str(nxt + 1)])
        prev = k
    row = [l.split() for l in open(gl) if l.split()[0] == gene][0]
    with open(f"{d}/all_{gene}_starting_kmers.txt", "wb") as f:
        step([BIN, "findstart", row[3], d + "/reads.lib.bin", "45", "16", f"{d}/35.contigs.fa"], stdout=f)
    lines = open(f"{d}/all_{gene}_starting_kmers.txt").read().splitlines()
    print(f"[{time.time() - t00:6.1f} s] {gene}: {len(lines)} seeds, the first {min(n_seeds, len(lines))} are searched", flush=True)
    open(f"{d}/t_{gene}_starting_kmers.txt", "w").write("\n".join(lines[:n_seeds]) + "\n")
    open(d + "/one_gene.txt", "w").write(" ".join(row) + "\n")
    for i, tr in enumerate(trials):
        env = {**os.environ, "MGTA_ASTAR_VERBOSE": "1", "MGTA_ASTAR_MONITOR": "30", **tr}
        t = time.time()
        try:
            r = subprocess.run([BIN, "search", f"{d}/44", d + "/one_gene.txt", d + "/t", f"{d}/out{i}", "20", "0.5", "16"], stdout=subprocess.DEVNULL, stderr=subprocess.PIPE,
                               text=True, env=env, timeout=limit)
            err, state = r.stderr, f"rc {r.returncode}"
        except subprocess.TimeoutExpired as e:
            err, state = (e.stderr.decode(errors="replace") if isinstance(e.stderr, bytes) else (e.stderr or "")), f"cut off at {limit} s"
        with open(os.path.join(logdir, f"trial_{n // 1_000_000}M_{gene}_{i}.log"), "w") as f:
            f.write(f"# {tr}\n" + err)
        tail = [l for l in err.splitlines() if "Done " in l or "[astar]" in l][-3:]
        print(f"[{time.time() - t00:6.1f} s] trial {i} {tr}: {state}, {time.time() - t:.1f} s\n    " + "\n    ".join(x[:420] for x in tail), flush=True)
finally:
    shutil.rmtree(d, ignore_errors=True)
