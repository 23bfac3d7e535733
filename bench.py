#!/usr/bin/env python3
"""bench.py — MegaGTA hot path on MI355X: SdBG build (Gk-mer/s) [+ A* expansions/s once built].

    python bench.py [--gpus N] [--steps K] [--warmup W] [--reads R] [--k 45]

A "step" = one pass of the hot path over one batch of synthetic input that is already resident in
HBM: packed reads -> SdBG edge stream (count, key generation, radix sort, edge emission) on every
rank's share of the 65536 prefix buckets; for N > 1 the record shards are all-gathered over RCCL so
every rank ends with the whole graph.  Work is fixed as N grows => "strong".
N = 1 workload = BASELINE.json configs[1]: rplB, 10 M x 150 bp reads, CLI k = 45 (graph k = 44).

One JSON line on rank 0: metric/value/unit per BASELINE.json, `roofline` for the dominant kernel
(radix scatter; algorithmic bytes / HIP-event duration measured inside the library on its own
stream) and `cpu_baseline` = the reference binary (oracle/_ref/megagta buildgraph, kind "reference")
or the oracle port, timed on this box's host cores on a bounded sample of the same reads.
"""
from __future__ import annotations

import argparse
import json
import os
import shutil
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak (MI355X_MICROARCH.md: 8 TB/s, ~6.3 TB/s achievable)


def b_build(k: int, L: int, edges_per_kmer: float) -> float:
    """algorithmic bytes per (k+1)-mer occurrence = 2 items x (write + read) x 4W + read the
    packed base once + 2 B per emitted edge."""
    W = (2 * k + 4 + 31) // 32
    return 2 * 2 * 4 * W + 0.25 * L / (L - k) + 2 * edges_per_kmer


def cpu_baseline(reads: np.ndarray, k: int, sample_reads: int) -> dict:
    from megagta_amd import synth
    n = min(sample_reads, reads.shape[0])
    sample = reads[:n]
    n_kmers = n * (reads.shape[1] - k)
    cores = os.cpu_count() or 1
    ref = os.path.join(ROOT, "oracle", "_ref", "megagta")
    tmp = tempfile.mkdtemp(prefix="mgta_cpu_")
    try:
        if os.path.exists(ref):
            synth.write_lib_bin(sample, os.path.join(tmp, "reads.lib"))
            threads = max(2, min(cores, 64))
            cmd = [ref, "buildgraph", "-k", str(k), "-m", "1", "--host_mem", str(32 << 30), "--mem_flag", "1", "--gpu_mem", "0",
                   "--output_prefix", os.path.join(tmp, "g"), "--num_cpu_threads", str(threads), "--num_output_threads", "1",
                   "--read_lib_file", os.path.join(tmp, "reads.lib")]
            t = time.time()
            subprocess.run(cmd, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
            dt = time.time() - t
            return {"value": n_kmers / dt / 1e9, "unit": "Gk-mer/s", "cores": threads, "kind": "reference",
                    "sample": f"first {n} reads x {reads.shape[1]} bp of the same set, graph k={k}, `megagta buildgraph` "
                              f"(reads.lib.bin -> .sdbg files, {dt:.2f} s wall incl. file I/O)"}
        from oracle import oracle as O
        packed, start = synth.pack_reads_for_build(sample)
        threads = min(cores, 32)
        t = time.time()
        O.Stream.build(packed, start, k, threads=threads)
        dt = time.time() - t
        return {"value": n_kmers / dt / 1e9, "unit": "Gk-mer/s", "cores": threads, "kind": "port",
                "sample": f"first {n} reads, graph k={k}, oracle restatement ({dt:.2f} s; key generation single-threaded)"}
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--reads", type=int, default=10_000_000)
    ap.add_argument("--k", type=int, default=45, help="CLI k (graph k = k-1, megagta.py:815-816)")
    ap.add_argument("--cpu-sample", type=int, default=1_000_000)
    ap.add_argument("--seeds", type=int, default=8000, help="seed k-mers of the A* leg (0 = skip the search leg)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    import torch
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    from megagta_amd import api, synth
    k = args.k - 1
    L = 150
    # identical synthetic read set on every rank (seeded); generated + packed in chunks on the host
    t0 = time.time()
    mg = synth.make_metagenome(args.reads, L, (("rplB", 277),), seed=1)
    packed, start = synth.pack_reads_for_build(mg.reads)
    t_gen = time.time() - t0

    ctx = api.Context(local_rank)
    rd = ctx.upload_reads(packed, start)            # inputs resident in HBM before the timed region
    from megagta_amd import dist as mdist
    b0, b1 = mdist.bucket_share(rank, world)

    def step():
        g = ctx.build_sdbg(rd, k, collect=False, bucket_range=(b0, b1))
        if world > 1:
            # the path's one exchange: every rank receives every shard of the edge stream (RCCL all-gather),
            # device to device: the shard never visits the host
            shard = api.export_records_to_torch(ctx)
            n = torch.tensor([shard.numel()], device="cuda", dtype=torch.int64)
            ns = [torch.zeros_like(n) for _ in range(world)]
            dist.all_gather(ns, n)
            mx = max(int(x.item()) for x in ns)
            pad = torch.zeros(max(mx, 1), dtype=torch.uint8, device="cuda")
            pad[: shard.numel()] = shard
            out = [torch.empty_like(pad) for _ in range(world)]
            dist.all_gather(out, pad)
        return g.stats

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    t = time.time()
    stats = []
    for _ in range(args.steps):
        stats.append(step())
    fence()
    dt = time.time() - t
    if world > 1:
        td = torch.tensor([dt], device="cuda", dtype=torch.float64)
        dist.all_reduce(td, op=dist.ReduceOp.MAX)
        dt = float(td.item())

    # ---- A* leg: graph + HMMs replicated, seeds dealt round-robin, one all-gather of contigs
    search = None
    findstart_leg = None
    denovo_leg = None
    if args.seeds > 0:
        import tempfile
        from megagta_amd import hmm as hmmlib
        if world > 1:
            g = ctx.build_sdbg(rd, k, collect=True, bucket_range=(b0, b1))
            graph = api.Graph(ctx, mdist.all_gather_edge_stream(g))
        else:
            ctx.build_sdbg(rd, k, collect=False)
            graph = api.Graph(ctx, None, k)                 # row f-4: the stream stays on the device between build and search
        td = tempfile.mkdtemp(prefix="mgta_bench_")
        synth.write_gene_models(mg.genes, td)
        fw = api.DeviceHmm(ctx, hmmlib.parse_hmm(os.path.join(td, "rplB", "for_enone.hmm")))
        rv = api.DeviceHmm(ctx, hmmlib.parse_hmm(os.path.join(td, "rplB", "rev_enone.hmm")))
        # seed finder (row f-2) on the reads already resident for the build: kernel time of one gene's scan
        from megagta_amd import findstart as fsm
        fwords, _ = fsm.reference_words(os.path.join(td, "rplB", "ref_aligned.faa"), args.k // 3)
        fhits, fms = fsm.find_hits(ctx, rd, True, args.k, fsm.pack_words(fwords, args.k // 3))
        fhits, fms = fsm.find_hits(ctx, rd, True, args.k, fsm.pack_words(fwords, args.k // 3))
        findstart_leg = {"ms_kernel": fms, "windows_per_s": args.reads * (L - args.k + 1) * 2 / (fms * 1e-3), "hits": int(fhits.size),
                         "reference_words": len(fwords), "note": "mgta_findstart, both strands, k=%d, one gene" % args.k}
        shutil.rmtree(td, ignore_errors=True)
        seeds = synth.synthetic_seeds(mg.genes[0], args.k, args.seeds, seed=4)
        mine = mdist.seed_share(len(seeds), rank, world)
        kmers, states = [seeds[i][0] for i in mine], [seeds[i][1] - 1 for i in mine]

        def sstep():
            res, st = api.astar_search(graph, fw, rv, kmers, states, 20, 0.5)
            if world > 1:
                mdist.all_gather_contigs(len(seeds), mine, [r.contig(km) for r, km in zip(res, kmers)])
            return st

        sstep()
        fence()
        t = time.time()
        sst = [sstep() for _ in range(max(1, args.steps // 2))]
        fence()
        sdt = (time.time() - t) / len(sst)
        nexp = torch.tensor([float(sst[-1]["n_expansions"]), sdt], dtype=torch.float64, device="cuda")
        if world > 1:
            ne = nexp[:1].clone()
            dist.all_reduce(ne)
            tm = nexp[1:].clone()
            dist.all_reduce(tm, op=dist.ReduceOp.MAX)
            nexp = torch.cat([ne, tm])
        search = {"value": float(nexp[0]) / float(nexp[1]), "unit": "HMM-scored node expansions/s", "n_seeds": len(seeds),
                  "expansions_per_step": float(nexp[0]), "ms_per_step": float(nexp[1]) * 1e3, "cache_mode": "cold (every seed independent)",
                  "bytes_per_expansion_algorithmic": 510, "achieved_GBps": float(nexp[0]) * 510 / float(nexp[1]) / 1e9,
                  "ms_kernel": sst[-1]["ms_kernel"], "retries": sst[-1]["n_retries"]}
        if world == 1:
            # row f-1: tips, bubbles, unitigs on the same resident graph (last: it consumes the validity bits)
            _, dst = graph.denovo(150, False, k + 2)
            denovo_leg = {"edges": int(graph.size), "ms_tips": dst["ms_tips"], "ms_bubbles": dst["ms_bubbles"], "ms_unitigs": dst["ms_unitigs"],
                          "tips": dst["n_tips"], "bubbles": dst["n_bubbles"], "bubble_rounds": dst["n_bubble_rounds"], "contigs": dst["n_contigs"],
                          "edges_per_s": graph.size / max(1e-9, (dst["ms_tips"] + dst["ms_bubbles"] + dst["ms_unitigs"]) * 1e-3),
                          "note": "mgta_denovo --max_tip_len 150 on the build leg's graph (one-thread-reference result, computed on the device)"}

    if rank == 0:
        s = stats[-1]
        n_kmers = s["n_kmers"]                       # every rank scans all reads: whole-job k-mers per step
        ms_step = dt / args.steps * 1e3
        value = n_kmers / (ms_step * 1e-3) / 1e9
        # the two big kernels of the build: radix_scatter (P launches: the most significant bytes) and local_sort (one launch:
        # every remaining digit in LDS).  Each launch reads and writes every key of this rank once = 16W bytes per
        # (k+1)-mer occurrence.  Durations: HIP events recorded by the library on its own stream.
        W = s["words_per_key"]
        items_per_launch = s["n_items"] / max(1, s["n_passes"])
        alg_bytes = items_per_launch * 4 * W * 2
        launches = sum(x["n_sort_launches"] for x in stats)
        ms_scatter = sum(x["ms_sort_scatter"] for x in stats) / max(1, launches)
        ms_local = sum(x["ms_local_sort"] for x in stats) / max(1, sum(x["n_passes"] for x in stats))
        scatter_total = sum(x["ms_sort_scatter"] for x in stats)
        local_total = sum(x["ms_local_sort"] for x in stats)
        dom_name, dom_ms = ("local_sort_kernel", ms_local) if local_total >= scatter_total else ("radix_scatter_kernel", ms_scatter)
        achieved = alg_bytes / (dom_ms * 1e-3) / 1e9 if dom_ms > 0 else 0.0
        traffic = None
        tp = os.path.join(ROOT, "profiles", "traffic_latest.json")
        if os.path.exists(tp):
            traffic = json.load(open(tp)).get(dom_name + "_bytes_per_launch")
        edges_per_kmer = s["n_edges"] / max(1, n_kmers) * world
        out = {
            "metric": "HMM-scored node expansions/sec + SdBG-build Gk-mer/s, k=45, 100Mx150bp",
            "value": value, "unit": "Gk-mer/s (SdBG build)", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_step, "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "u32",
            "data": "synthetic",
            "config": {"workload": f"rplB, {args.reads} x {L}bp synthetic reads, CLI k={args.k} (graph k={k}), -c 1, "
                                   f"{'bucket-range sharded, all-gather of record shards' if world > 1 else '1x MI355X'}",
                       "reads": args.reads, "read_len": L, "graph_k": k, "n_kmers": n_kmers, "n_items": s["n_items"],
                       "n_edges_rank0": s["n_edges"], "passes": s["n_passes"]},
            "roofline": {"bound": "hbm", "kernel": dom_name, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "avg_launch_ms": dom_ms,
                         "algorithmic_bytes_per_launch": alg_bytes,
                         "other_kernels": {"radix_scatter_kernel": {"avg_launch_ms": ms_scatter, "launches_per_step": launches / args.steps,
                                                                    "achieved": alg_bytes / (ms_scatter * 1e-3) / 1e9 if ms_scatter > 0 else 0.0},
                                           "local_sort_kernel": {"avg_launch_ms": ms_local, "launches_per_step": 1,
                                                                 "achieved": alg_bytes / (ms_local * 1e-3) / 1e9 if ms_local > 0 else 0.0}}},
            "whole_build": {"algorithmic_bytes_per_kmer": b_build(k, L, edges_per_kmer),
                            "achieved_GBps": n_kmers * b_build(k, L, edges_per_kmer) / (ms_step * 1e-3) / 1e9,
                            "phase_ms": {p: s[p] for p in ("ms_count", "ms_gen", "ms_sort", "ms_emit", "ms_total")},
                            "oversized_segments": s["n_big_segments"]},
            "host_prep_s": t_gen,
        }
        if search is not None:
            out["search"] = search
            out["findstart"] = findstart_leg
            if denovo_leg is not None:
                out["denovo"] = denovo_leg
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(mg.reads, k, args.cpu_sample)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
