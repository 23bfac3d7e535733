#!/usr/bin/env python3
"""`megagta search` over several GPUs of one node: one process per GPU, seeds sharded by gene first, ONE all-gather of contigs.

Same positional arguments, inputs and outputs as `megagta search` (search.cpp:72-90, megagta.py:682-684):

    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \\
        megagta_amd/search_dist.py <sdbg_prefix> <gene_list> <starting_kmers_prefix> <output_prefix> <prune_len> <low_cov_penalty> [threads]

(the driver does this for `megagta.py --gpus N`; with N = 1 it keeps the plain `megagta search`).  What replaces the reference's OpenMP
loop over seeds (search.cpp:184-189) across GPUs:
  * the graph (`<sdbg_prefix>.sdbg.*`) and the gene's two HMMs are replicated: every rank copies the files to its own GPU and parses the
    records there (`mgta_sdbg_load_files`; seconds for a graph of 630 M edges, no Python work per record);
  * seeds shard by GENE first, then round-robin inside a gene (`dist.gene_seed_share`): with N >= #genes every rank works on one gene;
  * every rank runs its seeds with the ordered-commit window over ITS sub-sequence of the seeds (MEGAGTA_CACHE_WINDOW /
    MEGAGTA_CACHE_COST_RATE, read exactly as `megagta search` reads them; defaults: window 1024 .. 8192 and cost term 4 or 2 by the number
    of the rank's seeds):
    seed j of a rank sees the paths of that rank's seeds <= j - B.  The result is a function of (seed order, N, B), never of timing;
    N = 1 is exactly `megagta search`;
  * ONE all-gather of the contig bytes per gene (RCCL over xGMI; gloo in the CPU tests), then rank 0 writes
    `<output_prefix>_raw_contigs_<gene>.fasta` in seed order with the reference's record names (hmm_graph_search.h:79).
No collective runs inside any kernel and none is needed before the end: the searches of different ranks share nothing.
"""
from __future__ import annotations

import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def read_gene_list(path: str) -> list[tuple[str, str, str]]:
    """name fwd.hmm rev.hmm [ref_aligned.faa] per line (search.cpp:105-122)"""
    out = []
    with open(path) as f:
        for line in f:
            a = line.split()
            if len(a) >= 3:
                out.append((a[0], a[1], a[2]))
    return out


def read_seeds(path: str) -> tuple[list[str], list[int]] | None:
    """8 whitespace-separated columns; column 4 = k-mer, column 8 = 1-based model position (search.cpp:149-158)"""
    if not os.path.exists(path):
        return None
    kmers, states = [], []
    with open(path) as f:
        for line in f:
            a = line.split()
            if len(a) >= 8:
                kmers.append(a[3])
                states.append(int(a[7]) - 1)
    return kmers, states


def write_fasta(path: str, gene: str, contigs, offsets) -> None:
    """`<output_prefix>_raw_contigs_<gene>.fasta` with the reference's record names (hmm_graph_search.h:79), contigs in seed order"""
    mv = memoryview(contigs)
    g = gene.encode()
    with open(path, "wb", buffering=1 << 22) as f:
        for i in range(len(offsets) - 1):
            f.write(b">%s_contig_%d_contig_%d\n" % (g, 2 * i, 2 * i + 1))
            f.write(mv[offsets[i]:offsets[i + 1]])
            f.write(b"\n")


DEFAULT_COST_KNEE, DEFAULT_COST_RATE2 = 0, 0      # (kDefaultCostKnee / kDefaultCostRate2 of csrc/host/megagta_main.cpp)


def search_plan(n_seeds: int) -> tuple[int, int, int, int]:
    """as `megagta search` (csrc/host/megagta_main.cpp::search_plan): the ordered-commit window and the cost term (rate, knee, rate beyond
    the knee; knee 0 = one rate) by the number of seeds the batch holds; MEGAGTA_CACHE_WINDOW / MEGAGTA_CACHE_COST_RATE /
    MEGAGTA_CACHE_COST_KNEE / MEGAGTA_CACHE_COST_RATE2 override (any integer >= -64 for the rate: < 0 = seeds per expansion)"""
    w, r = _env_int("MEGAGTA_CACHE_WINDOW"), _env_int("MEGAGTA_CACHE_COST_RATE")
    window = w if w is not None and w >= -1 else 1024 if n_seeds < 32768 else 2048 if n_seeds < 65536 else 4096 if n_seeds < 196608 else 8192
    rate = r if r is not None else 0 if window == 1 else (4 if n_seeds < 65536 else 2 if n_seeds < 393216 else 1)     # window 1 = the sequential run: no cost term
    k, r2 = _env_int("MEGAGTA_CACHE_COST_KNEE"), _env_int("MEGAGTA_CACHE_COST_RATE2")
    knee = max(0, k) if k is not None else DEFAULT_COST_KNEE
    rate2 = r2 if r2 is not None else DEFAULT_COST_RATE2
    if rate < 1 or knee <= 0 or rate2 <= rate:
        knee, rate2 = 0, 0
    return window, rate, knee, rate2


def window_and_rate(n_seeds: int):
    """(window, cost) for api.astar_search: cost = the rate, or (rate, knee, rate2) when the plan has a knee"""
    window, rate, knee, rate2 = search_plan(n_seeds)
    return window, ((rate, knee, rate2) if knee else rate)


def _env_int(name: str):
    """an integer from the environment exactly as `megagta search` reads it (env_int_strict, strtol's syntax): unset or empty = not given,
    anything that is not an integer is refused -- one environment means one mode for the binary and for the ranks"""
    import re
    v = os.environ.get(name)
    if v is None or v == "":
        return None
    if not re.fullmatch(r"[ \t\n\v\f\r]*[+-]?[0-9]+", v):
        raise SystemExit(f"{name} must be an integer (got '{v}')")
    return max(-1000000000, min(1000000000, int(v)))


def main(argv: list[str]) -> int:
    if len(argv) < 7:
        print(f"Usage: {argv[0]} <succinct_dbg> <gene_list> <starting_kmers_prefix> <output_prefix> <prune_len> <low_cov_penalty> [num_threads=0]",
              file=sys.stderr)
        return 1
    sdbg_prefix, gene_list, seeds_prefix, out_prefix = argv[1:5]
    prune, pen = int(argv[5]), float(argv[6])
    rank, world, local = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("LOCAL_RANK", "0"))
    import torch
    import torch.distributed as dist
    from megagta_amd import api, dist as mdist, hmm as hmmlib
    # MEGAGTA_DIST_BACKEND=gloo + MEGAGTA_DEVICE=0: several ranks on ONE GPU (tests on a one-GPU box; RCCL refuses two ranks per device)
    backend = os.environ.get("MEGAGTA_DIST_BACKEND", "nccl")
    device = int(os.environ.get("MEGAGTA_DEVICE", str(local)))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend == "nccl":
            torch.cuda.set_device(device)
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", device))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    t0 = time.time()
    ctx = api.Context(device)
    graph = api.Graph.from_files(ctx, sdbg_prefix)
    if rank == 0:
        print(f"    [megagta_amd] rank 0 of {world}: graph of {graph.size} edges on the device ({time.time() - t0:.2f} s)", file=sys.stderr, flush=True)
    if rank == 0 and world > 1 and search_plan(1 << 20)[0] != 0:
        # (measured: one gene split over two ranks differs from the one-rank run on ~1 % of its seeds, tests/test_multi_gpu_gpu.py)
        print(f"    [megagta_amd] note: {world} ranks -- every rank shares paths among the seeds IT searches (ordered window over its sub-sequence of the "
              f"seed list), so which of several equally good paths a seed takes is a function of (seed order, number of ranks): deterministic for a given "
              f"--gpus, not identical across rank counts.  MEGAGTA_CACHE_WINDOW=0 (no sharing) is independent of the rank count.", file=sys.stderr, flush=True)
    genes = read_gene_list(gene_list)
    seeds = [read_seeds(f"{seeds_prefix}_{name}_starting_kmers.txt") for name, _, _ in genes]
    share = mdist.gene_seed_share([len(s[0]) if s else 0 for s in seeds], rank, world)
    # every rank searches its share of every gene; the results stay on the rank until ONE all-gather at the end brings all genes' contigs
    # together (north_star: "a single RCCL all-gather of contigs over xGMI at the end"); rank 0 then writes the files
    results = []
    for gi, (name, fwd, rev) in enumerate(genes):
        if seeds[gi] is None:                                         # search.cpp:163-167: report and go on with the next gene
            if rank == 0:
                print(f"    [ERROR] Fail to open {seeds_prefix}_{name}_starting_kmers.txt", file=sys.stderr)
                open(f"{out_prefix}_raw_contigs_{name}.fasta", "w").close()
            continue
        kmers, states = seeds[gi]
        mine = share[gi]
        tg = time.time()
        contigs, offsets, nexp = np.zeros(0, np.uint8), np.zeros(1, np.int64), 0
        if mine.size:
            fw, rv = api.DeviceHmm(ctx, hmmlib.parse_hmm(fwd)), api.DeviceHmm(ctx, hmmlib.parse_hmm(rev))
            window, rate = window_and_rate(int(mine.size))            # per rank: over its own sub-sequence of the seeds
            contigs, offsets, st = api.astar_search_packed(graph, fw, rv, [kmers[i] for i in mine], [states[i] for i in mine], prune, pen,
                                                           cache_mode=window, cost_rate=rate)
            nexp = st["n_expansions"]
            fw.free(); rv.free()
        if world == 1:                                                # one rank: a gene's file is written as soon as the gene is searched
            write_fasta(f"{out_prefix}_raw_contigs_{name}.fasta", name, contigs, offsets)
            results.append((name, len(kmers), mine, None, None))
        else:
            results.append((name, len(kmers), mine, contigs, offsets))
        if rank == 0:
            print(f"    [megagta_amd] Done {name}: {len(kmers)} seeds over {world} rank(s), rank 0: {mine.size} seeds, {nexp} expansions, "
                  f"{time.time() - tg:.2f} s", file=sys.stderr, flush=True)
    tg = time.time()
    # the searches are over: the graph and the searches' pool (sized to most of the free memory) go BEFORE the exchange, which then has the
    # device to itself -- its staging is (world + 1) pieces of 64 MB, the contigs themselves stay on the host (advisor r4: the gather used to
    # ask for world x longest bytes next to a pool that had left 20 % free, at the very end of a run whose results were not on disk yet)
    graph.free()
    ctx.release_scratch()
    if world > 1:
        merged = mdist.all_gather_all_genes([r[1] for r in results], [r[2] for r in results], [r[3] for r in results], [r[4] for r in results])
        if rank == 0:
            for (name, _, _, _, _), (contigs, offsets) in zip(results, merged):
                write_fasta(f"{out_prefix}_raw_contigs_{name}.fasta", name, contigs, offsets)
            print(f"    [megagta_amd] {len(results)} gene(s): one all-gather of {sum(int(o[-1]) for _, o in merged)} contig bytes + the files in {time.time() - tg:.2f} s",
                  file=sys.stderr, flush=True)
    ctx.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(main(sys.argv))
