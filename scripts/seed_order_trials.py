#!/usr/bin/env python3
"""The search step of a multi-k run under several SEED ORDERS and sharing rules, on ONE set of artefacts (the pipeline up to the seeds runs
once, one process per step).  The reference's `findstart` shuffles its seed lines (fast_kmer_filter.cpp:183), so the order of the file is
the implementation's to choose; what `search` does with it (seed i sees the paths of the seeds before it, hmm_graph_search.h:279) makes the
order decide how much of every gene copy is explored cold.

python scripts/seed_order_trials.py <n_reads> <genes: rplB,nirK> <seconds per trial> "ORDER=midout,NAME=VAL;ORDER=lex;..." [log dir]
ORDER: lex (the file as `megagta findstart` writes it), midout (|model position - centre| ascending), desc / asc (model position),
       rand (seeded shuffle); every other NAME=VAL goes into the environment of `megagta search`."""
import os, shutil, subprocess, sys, tempfile, time, re
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from megagta_amd import synth

n = int(sys.argv[1])
genes = sys.argv[2].split(",")
limit = int(sys.argv[3])
trials = [dict(kv.split("=") for kv in t.split(",") if kv) for t in sys.argv[4].split(";") if t]
logdir = sys.argv[5] if len(sys.argv) > 5 else os.path.join(ROOT, "gpurun_out")
os.makedirs(logdir, exist_ok=True)
BIN = os.path.join(ROOT, "megagta_amd", "bin", "megagta")
d = tempfile.mkdtemp(prefix="mgta_order_")
t00 = time.time()


def step(cmd, stdout=None, env=None):
    t = time.time()
    r = subprocess.run(cmd, stdout=stdout if stdout else subprocess.DEVNULL, stderr=subprocess.PIPE, text=True, env=env)
    if r.returncode != 0:
        sys.exit(f"{cmd[1]} failed: {r.stderr[-1500:]}")
    print(f"[{time.time() - t00:6.1f} s] {cmd[1]} {time.time() - t:.1f} s", flush=True)
    return r


def reorder(lines, order, seed=7):
    pos = np.array([int(l.split("\t")[7]) for l in lines], dtype=np.int64)
    idx = np.arange(len(lines))
    if order == "lex":
        return lines
    if order == "rand":
        return [lines[i] for i in np.random.default_rng(seed).permutation(len(lines))]
    if order == "desc":
        o = np.lexsort((idx, -pos))
    elif order == "asc":
        o = np.lexsort((idx, pos))
    elif order == "midout":
        c = (int(pos.min()) + int(pos.max())) // 2
        o = np.lexsort((idx, np.abs(pos - c)))
    else:
        sys.exit(f"unknown ORDER {order}")
    return [lines[i] for i in o]


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
    rows = {l.split()[0]: l.split() for l in open(gl) if l.split()}
    seeds = {}
    for gene in genes:
        with open(f"{d}/all_{gene}_starting_kmers.txt", "wb") as f:
            step([BIN, "findstart", rows[gene][3], d + "/reads.lib.bin", "45", "16", f"{d}/35.contigs.fa"], stdout=f)
        seeds[gene] = open(f"{d}/all_{gene}_starting_kmers.txt").read().splitlines()
        print(f"[{time.time() - t00:6.1f} s] {gene}: {len(seeds[gene])} seeds", flush=True)
    open(d + "/genes.txt", "w").write("".join(" ".join(rows[g]) + "\n" for g in genes))
    base = {}
    for i, tr in enumerate(trials):
        tr = dict(tr)
        order = tr.pop("ORDER", "lex")
        for gene in genes:
            open(f"{d}/t_{gene}_starting_kmers.txt", "w").write("\n".join(reorder(seeds[gene], order)) + "\n")
        env = {**os.environ, "MGTA_ASTAR_VERBOSE": "1", **tr}
        t = time.time()
        try:
            r = subprocess.run([BIN, "search", f"{d}/44", d + "/genes.txt", d + "/t", f"{d}/out{i}", "20", "0.5", "16"], stdout=subprocess.DEVNULL, stderr=subprocess.PIPE,
                               text=True, env=env, timeout=limit)
            err, state = r.stderr, f"rc {r.returncode}"
        except subprocess.TimeoutExpired as e:
            err, state = (e.stderr.decode(errors="replace") if isinstance(e.stderr, bytes) else (e.stderr or "")), f"cut off at {limit} s"
        with open(os.path.join(logdir, f"order_{n // 1_000_000}M_{i}.log"), "w") as f:
            f.write(f"# ORDER={order} {tr}\n" + err)
        done = [l for l in err.splitlines() if "Done " in l]
        summ = []
        for l in done:
            m = re.search(r"Done (\S+): time ([0-9.]+) \((\d+) expansions.*largest search (\d+) nodes / (\d+) expansions", l)
            if m:
                summ.append(f"{m.group(1)} {float(m.group(2)):.1f} s {int(m.group(3)) / 1e6:.0f} M exp (largest {int(m.group(5)) / 1e3:.0f} k)")
        # the contigs as a multiset against the first trial's (orders differ, so seed indices do)
        same = ""
        for gene in genes:
            p = f"{d}/out{i}_raw_contigs_{gene}.fasta"
            if os.path.exists(p):
                cs = sorted(l for l in open(p) if not l.startswith(">"))
                if gene not in base:
                    base[gene] = cs
                else:
                    from collections import Counter
                    a, b = Counter(cs), Counter(base[gene])
                    same += f" {gene}: {sum((a & b).values())}/{len(cs)} contigs as in trial 0;"
        print(f"[{time.time() - t00:6.1f} s] trial {i} ORDER={order} {tr}: {state}, {time.time() - t:.1f} s | " + " | ".join(summ) + " |" + same, flush=True)
finally:
    shutil.rmtree(d, ignore_errors=True)
