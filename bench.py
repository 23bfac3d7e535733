#!/usr/bin/env python3
"""bench.py — MegaGTA hot path on MI355X: SdBG build (Gk-mer/s) + A* expansions/s + reads->contigs wall.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--reads R] [--k 45]

N = 1 workload = BASELINE.json configs[2], the configuration the metric is quoted on: rplB + nirK, 100 M x 150 bp synthetic reads,
CLI k = 45 (graph k = 44); `--reads 10000000` is configs[1].  The reads are generated and packed ON the device (seeded; every rank
draws the same set) and are resident in HBM before the timed region starts.

A "step" = one pass of the hot path over that input: packed reads -> SdBG edge stream (count, key generation, radix sort, edge
emission) on every rank's share of the 65536 prefix buckets; for N > 1 the record shards are all-gathered over RCCL, device to device,
so every rank ends with the whole stream.  Work is fixed as N grows => "strong".

One JSON line on rank 0: metric / value / unit per BASELINE.json;
  `roofline`     the dominant kernel of the build (algorithmic bytes / HIP-event duration measured inside the library on its own stream);
  `search`       the A* leg on the graph of that build (graph + HMMs replicated; seeds shard by gene, then round-robin; one all-gather
                 of contigs): expansions/s, achieved bytes/s against the random-line peak the library's own HIP probe measures in the same run;
  `e2e`          reads.fa -> contigs through the driver (`megagta.py -k 30,36,45`, rplB + nirK) on a bounded sample (2 M reads), next to the
                 reference binary behind the same driver on the SAME files (`speedup_same_sample`);
  `parity_1M`    the device's edge stream of the CPU-baseline sample against the reference binary's .sdbg files of it (bit-exact or not);
  `cpu_baseline` the reference `buildgraph` (kind "reference") or the oracle port on this box's host cores, bounded sample.
"""
from __future__ import annotations

import argparse
import collections
import hashlib
import json
import os
import shutil
import signal
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak (MI355X_MICROARCH.md: 8 TB/s, ~6.3 TB/s achievable)
REF = os.path.join(ROOT, "oracle", "_ref", "megagta")
DRIVER = os.path.join(ROOT, "megagta_amd", "megagta.py")
RECORDED_REFERENCE = os.path.join(ROOT, "profiles", "r06", "e2e_10M_reference.json")   # scripts/e2e_reference_at_size.py: ours and the reference on 10 M reads


def host_cores() -> dict:
    """logical CPUs this process may use and the box's physical cores ((physical id, core id) pairs of /proc/cpuinfo)"""
    logical = os.cpu_count() or 1
    try:
        usable = len(os.sched_getaffinity(0))
    except AttributeError:
        usable = logical
    phys = set()
    try:
        pid = cid = None
        for line in open("/proc/cpuinfo"):
            if line.startswith("physical id"):
                pid = line.split(":")[1].strip()
            elif line.startswith("core id"):
                cid = line.split(":")[1].strip()
            elif not line.strip():
                if pid is not None and cid is not None:
                    phys.add((pid, cid))
                pid = cid = None
    except OSError:
        pass
    return {"logical_cpus": logical, "usable_cpus": usable, "physical_cores": len(phys) or None}


def source_signature(*names: str) -> str:
    """md5 over the kernel sources a profile was taken with: a PMC summary under profiles/ is only quoted in the bench line while the
    kernels it measured are the ones that ran (a changed kernel makes the file stale, and the line says so instead of quoting it)"""
    import hashlib
    h = hashlib.md5()
    for n in names:
        with open(os.path.join(ROOT, "megagta_amd", "csrc", n), "rb") as f:
            h.update(f.read())
    return h.hexdigest()


BUILD_SOURCES = ("sdbg_build.hip", "sdbg_solid.hpp", "scan.hpp", "device_utils.hpp")
ASTAR_SOURCES = ("astar_kernel.hpp", "graph.hpp", "device_utils.hpp")


def b_build(k: int, L: int, edges_per_kmer: float) -> float:
    """algorithmic bytes per (k+1)-mer occurrence = 2 items x (write + read) x 4W + read the
    packed base once + 2 B per emitted edge."""
    W = (2 * k + 4 + 31) // 32
    return 2 * 2 * 4 * W + 0.25 * L / (L - k) + 2 * edges_per_kmer


def parity_vs_reference_graph(ctx, sample: np.ndarray, k: int, ref_prefix: str) -> dict:
    """the reference's `buildgraph` output of the CPU-baseline sample (its .sdbg files, decoded by the oracle's reader --
    the checker, outside every timed region) against the edge stream the device builds from the same reads: bit-exact or not."""
    from megagta_amd import synth
    from oracle import oracle as O
    packed, start = synth.pack_reads_for_build(sample)
    rd = ctx.upload_reads(packed, start)
    t = time.time()
    ours = ctx.build_sdbg(rd, k)
    dt = time.time() - t
    rd.free()
    ref = O.Stream.read(ref_prefix).edges()
    return {"reads": int(sample.shape[0]), "edges": int(ours.records.size), "edges_reference": int(ref.records.size), "tips": int(ours.tips.size // max(1, ours.words_per_tip)),
            "md5_equal": ours.md5() == ref.md5(), "md5": ours.md5(), "device_build_ms": ours.stats["ms_total"], "device_build_and_copy_s": dt,
            "note": "edge stream (bucket sizes, records, large multiplicities, tip labels) of the CPU-baseline sample: device vs the reference binary's .sdbg files"}


def search_cpu_baseline(tmp: str, graph_prefix: str, lib_bin: str, genes, k: int, cores: int, n_seeds: int = 8000) -> dict:
    """The search half of the metric on the host: the reference `search` (search.cpp:71-197, OMP over the seeds) on the graph it has just
    built of the CPU-baseline sample, seeds of the first gene from its own `findstart` (a contiguous block of the sorted list, as the
    product-mode leg takes them), best of 12 / 16 / 32 threads (each run on its own: one that crashes costs its own number only); seconds = its own "Done <gene>: time" line (the seed loop, search.cpp:184-194).
    Expansions = closed-set insertions + one start expansion per search, counted by the reference's own classes run
    sequentially over the same seeds (oracle/_ref/probe astar ... warm: the multi-thread run shares its caches by timing, so its own
    count differs from run to run by a little; it prints none)."""
    import re
    from megagta_amd import synth
    probe = os.path.join(ROOT, "oracle", "_ref", "probe")
    gl = synth.write_gene_models(genes[:1], os.path.join(tmp, "models"))
    name, fwd, rev, faa = open(gl).readline().split()
    lines = sorted(subprocess.run([REF, "findstart", faa, lib_bin, str(k + 1), str(min(cores, 16))], check=True, capture_output=True).stdout.decode().splitlines())
    lo = max(0, (len(lines) - n_seeds) // 2)
    lines = lines[lo:lo + n_seeds]
    if not lines:
        return {"error": "the reference's findstart found no seed in the sample"}
    sp = os.path.join(tmp, "sb")
    open(f"{sp}_{name}_starting_kmers.txt", "w").write("\n".join(lines) + "\n")
    # One run per thread count, each on its own: the reference's `term_nodes.find` is unlocked and races with a rehash of the shared table
    # (hmm_graph_search.h:279, hash_table_st.h:554-568), so a run can die (SIGSEGV at 64 threads in round 4) -- the counts that survive are
    # kept, the ones that crashed are named.  12 = the cap of the reference's own driver (megagta.py:683); 64 threads are not tried any more.
    secs, crashed = {}, {}
    for threads in sorted({max(1, min(cores, 12)), max(1, min(cores, 16)), max(1, min(cores, 32))}):
        for attempt in range(2):                                         # (a crashed count is tried once more: the race is a matter of timing)
            t = time.time()
            try:
                r = subprocess.run([REF, "search", graph_prefix, gl, sp, os.path.join(tmp, f"so{threads}"), "20", "0.5", str(threads)], capture_output=True, timeout=120)
            except subprocess.TimeoutExpired:
                crashed[threads] = "no end after 120 s"
                break
            wall = time.time() - t
            if r.returncode != 0:
                crashed[threads] = f"exit status {r.returncode}" + (" (signal %d)" % -r.returncode if r.returncode < 0 else "")
                continue
            crashed.pop(threads, None)
            m = re.search(r"Done %s: time ([0-9.]+)" % re.escape(name), r.stderr.decode(errors="replace"))
            secs[threads] = float(m.group(1)) if m else wall
            break
    if not secs:
        return {"error": "the reference's `search` ended abnormally at every thread count", "crashed_by_threads": crashed}
    best_t = min(secs, key=secs.get)
    t = time.time()
    pr = subprocess.run([probe, "astar", graph_prefix, fwd, rev, f"{sp}_{name}_starting_kmers.txt", "20", "0.5", "warm"], check=True, capture_output=True).stdout.decode()
    t_probe = time.time() - t
    closed = sum(int(x) for x in re.findall(r" closed (\d+) ", pr))
    n_exp = closed + 2 * len(lines)
    return {"value": n_exp / secs[best_t], "unit": "HMM-scored node expansions/s", "cores": best_t, "kind": "reference",
            "seeds": len(lines), "gene": name, "expansions": n_exp, "seconds": secs[best_t], "seconds_by_threads": secs, "crashed_by_threads": crashed,
            "one_thread_sequential": {"seconds": t_probe, "value": n_exp / t_probe, "note": "oracle/_ref/probe astar warm: the reference's classes, one thread, incl. loading the graph"},
            "sample": f"{len(lines)} `findstart` seeds of {name} (a contiguous block of the sorted list) on the reference's own graph of the first reads (graph k={k}); "
                      f"`megagta search ... {best_t}`, seed-loop seconds from its own log line; expansions counted by a sequential run of the same classes"}


def cpu_baseline(reads: np.ndarray, k: int, sample_reads: int, ctx=None, genes=None) -> dict:
    from megagta_amd import synth
    n = min(sample_reads, reads.shape[0])
    sample = reads[:n]
    n_kmers = n * (reads.shape[1] - k)
    cores = os.cpu_count() or 1
    tmp = tempfile.mkdtemp(prefix="mgta_cpu_")
    try:
        if os.path.exists(REF):
            synth.write_lib_bin(sample, os.path.join(tmp, "reads.lib"))
            best, failed = None, {}
            for threads in sorted({max(2, min(cores, 16)), max(2, min(cores, 64))}):     # the reference does not scale to every core: keep the best
                cmd = [REF, "buildgraph", "-k", str(k), "-m", "1", "--host_mem", str(32 << 30), "--mem_flag", "1", "--gpu_mem", "0",
                       "--output_prefix", os.path.join(tmp, f"g{threads}"), "--num_cpu_threads", str(threads), "--num_output_threads", "1",
                       "--read_lib_file", os.path.join(tmp, "reads.lib")]
                t = time.time()
                rb = subprocess.run(cmd, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
                dt = time.time() - t
                if rb.returncode != 0:                                   # (one thread count that fails costs its own number only)
                    failed[threads] = rb.returncode
                    continue
                if best is None or dt < best[0]:
                    best = (dt, threads)
            if best is None:
                return {"error": f"the reference's `buildgraph` failed at every thread count: {failed}"}
            dt, threads = best
            out = {"value": n_kmers / dt / 1e9, "unit": "Gk-mer/s", "cores": threads, "kind": "reference",
                   "sample": f"first {n} reads x {reads.shape[1]} bp of the same set, graph k={k}, `megagta buildgraph` "
                             f"(reads.lib.bin -> .sdbg files, {dt:.2f} s wall incl. file I/O; best of 16 / 64 threads)"}
            out["host"] = host_cores()
            if ctx is not None:
                try:
                    out["_parity"] = parity_vs_reference_graph(ctx, sample, k, os.path.join(tmp, f"g{threads}"))
                except Exception as e:                                   # the baseline number must not die with the check
                    out["_parity"] = {"error": str(e)[-400:]}
            if genes:
                try:
                    out["_search"] = search_cpu_baseline(tmp, os.path.join(tmp, f"g{threads}"), os.path.join(tmp, "reads.lib.bin"), genes, k, cores)
                except Exception as e:
                    out["_search"] = {"error": str(e)[-400:]}
            return out
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


def write_fasta_fast(reads: np.ndarray, path: str) -> None:
    """>r0000000\\nACGT...\\n per read, assembled as one byte matrix"""
    n, L = reads.shape
    width = 8
    rec = np.empty((n, 1 + 1 + width + 1 + L + 1), dtype=np.uint8)
    rec[:, 0] = ord(">")
    rec[:, 1] = ord("r")
    ids = np.arange(n, dtype=np.int64)
    for d in range(width):
        rec[:, 2 + width - 1 - d] = (ids % 10 + ord("0")).astype(np.uint8)
        ids //= 10
    rec[:, 2 + width] = ord("\n")
    rec[:, 3 + width:3 + width + L] = np.frombuffer(b"ACGT", dtype=np.uint8)[reads]
    rec[:, -1] = ord("\n")
    rec.tofile(path)


def e2e_leg(gene_specs, n_ours: int, n_ref: int, device: str, klist: str = "30,36,45", n_large: int = 0, large_deadline: float = 0.0,
            hard_stop: float = 0.0) -> dict:
    """reads.fa -> contigs/<gene>/{nucl,prot}_merged.fasta through megagta.py, wall seconds.  Two read sets of their own (the same
    model as the build leg's, 15x coverage each: a prefix of the 100 M reads would be a 0.3x sample of 50 000 genomes with next to
    nothing to assemble): `n_ours` reads for our driver run, and `n_ref` reads on which the reference binary runs behind the same driver
    (thread count swept, best kept) and ours runs too, for an equal-work ratio."""
    from megagta_amd import synth
    cores = os.cpu_count() or 1
    tmp = tempfile.mkdtemp(prefix="mgta_e2e_")
    out = {"k_list": klist, "genes": [g[0] for g in gene_specs]}
    try:
        sets = {}
        contents = {}                                                    # tag -> gene -> Counter of the contigs' md5 digests

        def equal_fraction(tag_a, tag_b):
            """per gene: contigs of run `tag_a` that run `tag_b` holds too, as multisets, over the contigs of `tag_b`"""
            out_ = {}
            for g, cb in contents[tag_b].items():
                ca = contents[tag_a].get(g, collections.Counter())
                nb = sum(cb.values())
                out_[g] = (sum((ca & cb).values()) / nb) if nb else (1.0 if not ca else 0.0)
            return out_

        def ensure_set(n):
            if n not in sets:
                mg = synth.make_metagenome_device(n, 150, gene_specs, seed=1000 + n % 997, device=device, host_sample=n)
                d = os.path.join(tmp, f"set_{n}")
                gl_ = synth.write_gene_models(mg.genes, os.path.join(d, "models"))
                write_fasta_fast(mg.sample_reads, os.path.join(d, "reads.fa"))
                sets[n] = (os.path.join(d, "reads.fa"), gl_, [g.name for g in mg.genes])
                del mg

        def run(n, tag, extra, env=None):
            ensure_set(n)
            fa, gl, names = sets[n]
            od = os.path.join(tmp, "out_" + tag)
            t = time.time()
            # (last resort: the driver's call of bench.py ends at 600 s and the CPU baselines come after this leg -- a run that would carry the
            # leg beyond `hard_stop` is ended, with its process group, and the line goes out without it)
            p = subprocess.Popen([sys.executable, DRIVER, "-r", fa, "-g", gl, "-k", klist, "-o", od, "-c", "1"] + extra,
                                 stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, text=True, env={**os.environ, **(env or {})}, start_new_session=True)
            try:
                _, err = p.communicate(timeout=max(5.0, hard_stop - time.time()) if hard_stop else None)
            except subprocess.TimeoutExpired:
                try:
                    os.killpg(p.pid, signal.SIGKILL)
                except ProcessLookupError:
                    pass
                p.communicate()
                shutil.rmtree(od, ignore_errors=True)
                raise TimeoutError(f"megagta.py ({tag}, {n} reads) was still running when the bench run was {time.time() - _T0:.0f} s old: ended, not measured")
            r = subprocess.CompletedProcess(p.args, p.returncode, None, err)
            dt = time.time() - t
            if os.environ.get("MEGAGTA_E2E_LOG_DIR"):                   # the driver's own log (per-step times), for profiles/
                with open(os.path.join(os.environ["MEGAGTA_E2E_LOG_DIR"], f"e2e_{tag}.log"), "w") as f:
                    f.write(r.stderr)
                    if os.path.exists(os.path.join(od, "log")):
                        f.write("\n==== <out>/log ====\n" + open(os.path.join(od, "log"), errors="replace").read())
            if r.returncode != 0:
                raise RuntimeError(f"megagta.py ({tag}) failed: {r.stderr[-800:]}")
            n_contigs, content = {}, {}
            for g in names:                                              # the CONTENT of the run's result, gene by gene: a multiset of sequence digests
                c = collections.Counter()
                with open(os.path.join(od, "contigs", g, "nucl_merged.fasta"), "rb") as f:
                    for l in f:
                        if not l.startswith(b">"):
                            c[hashlib.md5(l.strip().upper()).digest()] += 1
                n_contigs[g] = sum(c.values())
                content[g] = c
            if n <= 5_000_000 or os.environ.get("MEGAGTA_E2E_KEEP_CONTENT"):   # (the ours-only point beyond the same-sample size is compared with nothing)
                contents[tag] = content
            shutil.rmtree(od, ignore_errors=True)
            return dt, n_contigs

        dt, nc = run(n_ours, "ours", ["-t", str(min(cores, 16))])
        note(f"e2e ours: {n_ours} reads in {dt:.1f} s")
        out["ours"] = {"reads": n_ours, "seconds": dt, "reads_per_s": n_ours / dt, "contigs": nc,
                       "note": "default: shared term_nodes caches under the ordered-commit window (the same files on every run)"}
        if not os.environ.get("MEGAGTA_E2E_SKIP_UNORDERED"):           # (set for one-off runs at sizes where a second run does not fit the call)
            dtu, ncu = run(n_ours, "ours_unordered", ["-t", str(min(cores, 16))], env={"MEGAGTA_CACHE_WINDOW": "-1"})
            note(f"e2e ours, unordered cache sharing: {n_ours} reads in {dtu:.1f} s")
            out["ours_unordered_cache"] = {"reads": n_ours, "seconds": dtu, "reads_per_s": n_ours / dtu, "contigs": ncu,
                                           "note": "MEGAGTA_CACHE_WINDOW=-1: every search sees whatever paths are in the cache when it looks, as the "
                                                   "reference's multi-thread `search` does; which of several equally scored paths a seed takes depends on timing"}
        # The reference binary behind the same driver (CPU only) and our larger run (GPU + a few host threads) use different parts of the box: they
        # run SIDE BY SIDE (half of the driver's 443 s was the reference on the host while the GPU idled).  The reference has its
        # `-t` threads to itself as long as the host has cores for both (64 on the pool's boxes: 32 + ours' 16); MEGAGTA_E2E_SERIAL=1 runs them one
        # after the other as rounds 2-5 did.
        import threading
        ref_box = {}

        def reference_leg():
            # Its best thread count is found on the small set (16 / 32; every core was 3.7x slower than 32 in round 2, 64 threads 1.2x slower
            # in round 4), then it runs ONCE on the SAME files as ours above: one equal-work ratio, nothing else
            sweep = {}
            forced = os.environ.get("MEGAGTA_E2E_REF_THREADS")          # (one-off runs at sizes where the sweep does not fit the call)
            best_t = int(forced) if forced else 0
            try:
                crashed = {}
                for threads in ([] if forced else sorted({min(cores, 16), min(cores, 32)})):     # (64 threads and every core were slower than 32 in every sweep of rounds 2-4)
                    try:
                        dtr, _ = run(n_ref, f"ref_small_t{threads}", ["--bin", REF, "-t", str(threads)])
                    except RuntimeError as e:
                        # the reference's multi-thread `search` reads term_nodes without a lock (search.cpp:182-189) and segfaults now and then: that
                        # thread count is named and left out, the leg goes on
                        crashed[threads] = str(e)[-200:]
                        note(f"e2e reference, {threads} threads: the run on {n_ref} reads FAILED (left out of the sweep)")
                        continue
                    sweep[threads] = dtr
                    note(f"e2e reference, {threads} threads: {n_ref} reads in {dtr:.1f} s")
                if sweep or crashed:
                    ref_box["sweep"] = {"reads": n_ref, "seconds_by_threads": sweep, "crashed": crashed}
                if sweep:
                    best_t = min(sweep, key=sweep.get)
                elif not forced:
                    best_t = min(cores, 16)
                tries = [best_t] + [t for t in sorted(sweep, key=sweep.get) if t != best_t]
                last = None
                for t_ in tries:                                         # (a crash of the big run: once more with the next-best thread count)
                    try:
                        dtr, ncr = run(n_ours, f"ref_t{t_}", ["--bin", REF, "-t", str(t_)])
                        best_t = t_
                        last = None
                        break
                    except RuntimeError as e:
                        last = e
                        note(f"e2e reference, {t_} threads: the run on {n_ours} reads FAILED")
                if last is not None:
                    raise last
                note(f"e2e reference, {best_t} threads: {n_ours} reads in {dtr:.1f} s")
                ref_box.update(best_t=best_t, dtr=dtr, ncr=ncr)
            except TimeoutError as e:
                ref_box.update(best_t=best_t, cut_off=str(e))
            except Exception as e:                                       # noqa: BLE001 -- reported in the line, never lost in a thread
                ref_box.update(best_t=best_t, error=repr(e))

        def large_leg():
            # a point beyond the same-sample size, ours only (the reference needs ~60 s per million reads; profiles/r06 holds it at 10 M reads): skipped when the run is late
            if time.time() > large_deadline:
                out["ours_large"] = {"reads": n_large, "skipped": "the bench run was %.0f s old when this leg was due: not started" % (time.time() - _T0)}
                return
            try:
                dtl, ncl = run(n_large, "ours_large", ["-t", str(min(cores, 16))])
            except TimeoutError as e:
                out["ours_large"] = {"reads": n_large, "cut_off": str(e)}
                return
            note(f"e2e ours: {n_large} reads in {dtl:.1f} s")
            out["ours_large"] = {"reads": n_large, "seconds": dtl, "reads_per_s": n_large / dtl, "contigs": ncl,
                                 "note": "megagta.py -k %s on %d reads, default mode, ours only in THIS run" % (klist, n_large)}
            # the reference at this size is a quarter of an hour of host time: measured once by the builder on a box of the same pool, on the same
            # generated read set (same seed, same files) -- quoted with its date and commit, never re-timed here
            try:
                rj = json.load(open(RECORDED_REFERENCE))
                if rj["reference"]["reads"] == n_large and rj.get("k_list") == klist and rj.get("genes") == out["genes"] and "seconds" in rj["reference"]:
                    out["ours_large"]["speedup_vs_recorded_reference"] = {
                        "value": rj["reference"]["seconds"] / dtl, "reference_seconds": rj["reference"]["seconds"], "reference_threads": rj["reference"]["threads"],
                        "ours_seconds_in_that_run": rj["ours"]["seconds"], "speedup_in_that_run": rj.get("speedup_same_sample"),
                        "contigs_equal_fraction_in_that_run": rj.get("contigs_equal_fraction"), "recorded": rj.get("date"), "commit": rj.get("commit"),
                        "file": os.path.relpath(RECORDED_REFERENCE, ROOT), "note": "the reference was timed on another box of the pool (same image, same generated reads)"}
            except Exception:                                           # noqa: BLE001 -- no record: no ratio
                pass

        have_ref = n_ref > 0 and os.path.exists(REF)
        side_by_side = have_ref and n_large > 0 and not os.environ.get("MEGAGTA_E2E_SERIAL") and cores >= 48
        th = None
        if have_ref:
            if not os.environ.get("MEGAGTA_E2E_REF_THREADS"):
                ensure_set(n_ref)                                        # (the read sets are made on the GPU: before the threads part)
            if side_by_side:
                ensure_set(n_large)
                th = threading.Thread(target=reference_leg)
                th.start()
            else:
                reference_leg()
        try:
            if n_large > 0 and not ref_box.get("cut_off"):
                large_leg()
        finally:
            if th is not None:                                           # (never leave the leg -- and remove its files -- under a running reference)
                th.join()
        if have_ref:
            out["reference_ran"] = "beside the GPU's %d-read leg (host cores: %d; the reference's threads + ours' 16 fit)" % (n_large, cores) if side_by_side else "alone on the host"
            if "sweep" in ref_box:
                out["reference_thread_sweep"] = ref_box["sweep"]
            if "error" in ref_box:                                       # (ours' numbers stay in the line whatever the reference did)
                out["reference"] = {"reads": n_ours, "threads": ref_box.get("best_t"), "error": ref_box["error"][-600:]}
                return out
            if "cut_off" in ref_box:
                out["reference"] = {"reads": n_ours, "threads": ref_box.get("best_t"), "cut_off": ref_box["cut_off"]}
                return out
            best_t, dtr, ncr = ref_box["best_t"], ref_box["dtr"], ref_box["ncr"]
            out["reference"] = {"reads": n_ours, "seconds": dtr, "threads": best_t, "reads_per_s": n_ours / dtr, "contigs": ncr,
                                "note": "the reference binary behind the same driver on the SAME files as `ours`; thread count = the best of "
                                        "16 / 32 on the small set"}
            out["speedup_same_sample"] = dtr / dt
            # what the two runs WROTE, not how much: nucl_merged.fasta of ours against the reference's on the same reads, gene by gene, as
            # multisets of sequences.  Below 1 by construction: the reference's own result depends on its seed order (findstart shuffles,
            # fast_kmer_filter.cpp:183) and on the timing of its threads' cache sharing (search.cpp:182-189)
            out["contigs_equal_fraction"] = equal_fraction("ours", f"ref_t{best_t}")
            if "ours_unordered" in contents:
                out["contigs_equal_fraction_unordered_cache"] = equal_fraction("ours_unordered", f"ref_t{best_t}")
                out["contigs_equal_fraction_ours_ordered_vs_unordered"] = equal_fraction("ours_unordered", "ours")
            if "ours_unordered_cache" in out:
                out["speedup_same_sample_unordered_cache"] = dtr / out["ours_unordered_cache"]["seconds"]
        return out
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def product_seed_list(hits: np.ndarray, sample: np.ndarray, words, model_pos, k: int) -> list[tuple[str, int]]:
    """(k-mer, 1-based model position) of the hits that fall into the reads of `sample` (uint8 codes [n, L], as sequenced): unique by
    k-mer, sorted -- the lines `megagta findstart` would write for those reads (fast_kmer_filter.cpp:181-188), built with numpy"""
    n, L = sample.shape
    h = hits[hits["read"] < n]
    if h.size == 0:
        return []
    pos, strand = (h["pos_strand"] >> 1).astype(np.int64), (h["pos_strand"] & 1).astype(bool)
    ar = np.arange(k, dtype=np.int64)
    col = pos[:, None] + ar[None, :]
    col = np.where(strand[:, None], L - 1 - col, col)
    km = sample[h["read"].astype(np.int64)[:, None], col]
    km = np.where(strand[:, None], 3 - km, km).astype(np.uint8)
    uniq, first = np.unique(km, axis=0, return_index=True)             # lexicographic row order = the sorted file
    text = np.frombuffer(b"ACGT", dtype=np.uint8)[uniq]
    refs = h["ref"][first]
    return [(text[i].tobytes().decode(), int(model_pos[int(refs[i])])) for i in range(uniq.shape[0])]


def contig_membership(graph, last_contigs: dict, k: int, n_sample: int = 200_000) -> dict:
    """Every contig the search returns is a walk in the graph: a sample of the (k+1)-mers of the returned contigs (seed-only contigs
    left out: a seed no read covers is returned as it is) looked up with IndexBinarySearchEdge on the device.  At 100 M reads most edge
    ids lie beyond 2^32: the check that catches what only shows at that size (a launch of more than 2^32 work-items, 32-bit ids)."""
    rng = np.random.default_rng(5)
    rows = []
    for cont, offs in last_contigs.values():
        lens = np.diff(offs)
        ok = np.nonzero(lens > k + 1)[0]
        if ok.size == 0:
            continue
        c = rng.choice(ok, size=min(n_sample // max(1, len(last_contigs)), ok.size * 4))
        p = (rng.random(c.size) * (lens[c] - k)).astype(np.int64)
        rows.append(cont[(offs[c] + p)[:, None] + np.arange(k + 1)[None, :]])
    if not rows:
        return {"sampled": 0}
    km = np.concatenate(rows)
    code = np.zeros(256, np.uint8)
    for ch, v in ((b"a", 1), (b"c", 2), (b"g", 3), (b"t", 4), (b"A", 1), (b"C", 2), (b"G", 3), (b"T", 4)):
        code[ch[0]] = v
    seqs = np.ascontiguousarray(code[km])
    ids = np.empty(seqs.shape[0], dtype=np.int64)
    from megagta_amd._lib import check
    check(graph.ctx._L.mgta_sdbg_index_edges(graph.h, seqs.ctypes.data, seqs.shape[0], ids.ctypes.data), "mgta_sdbg_index_edges")
    return {"sampled": int(ids.size), "found": int((ids >= 0).sum()), "ids_above_2^32": int((ids >= 2 ** 32).sum()), "max_edge_id": int(ids.max())}


def random_line_probe(ctx, table_bytes: int = 8 << 30) -> dict:
    """What this device sustains on random 128-byte lines, measured by the library's own HIP probe (mgta_probe_random_lines, csrc/probe.hip)
    with the kernels' access shape -- groups of 8 lanes read one aligned line each -- over a table far larger than every cache:
    independent lines swept over lines in flight per CU (4 ... 2048) and the loaded latency of ONE dependent line (pointer chase).
    The A* leg and the graph walks under it are priced against these ("achieved random-sector rate vs a measured
    pointer-chase ceiling"); the torch gather of rounds 2-5 was not a ceiling (the kernel exceeded it)."""
    indep = []
    for lif in (4, 8, 16, 32, 64, 128, 256, 512, 1024, 2048):           # waves/CU x groups x unroll: groups first, then waves, then unroll
        g = min(8, lif); w = min(32 if lif > 1024 else 16, max(1, lif // g)); u = max(1, lif // (g * w))
        indep.append((w, g, u, 0))
    dep = [(1, 1, 1, 1), (8, 8, 1, 1), (16, 8, 1, 1), (32, 8, 1, 1)]    # idle chip; 64 / 128 / 256 chains per CU

    def run(cfgs, target_ms=30.0):
        first = ctx.probe_random_lines(table_bytes, [(*c, 128) for c in cfgs])           # a short pass sizes the timed one
        return ctx.probe_random_lines(table_bytes, [(*c, min(1 << 22, max(128, int(128 * target_ms / max(r["ms"], 1e-3))))) for c, r in zip(cfgs, first)])

    ri, rdep = run(indep), run(dep)
    best = max(ri, key=lambda r: r["gb_per_s"])
    return {"table_bytes": table_bytes, "access": "groups of 8 lanes x 16 B = one aligned 128-byte line per group",
            "independent": [{"lines_in_flight_per_cu": r["lines_in_flight_per_cu"], "gb_per_s": r["gb_per_s"]} for r in ri],
            "dependent": [{"chains_per_cu": r["lines_in_flight_per_cu"], "ns_per_line": r["ns_per_step"], "gb_per_s": r["gb_per_s"]} for r in rdep],
            "peak_gb_per_s": best["gb_per_s"], "peak_at_lines_in_flight_per_cu": best["lines_in_flight_per_cu"],
            "idle_dependent_ns": rdep[0]["ns_per_step"]}


_T0 = time.time()


def note(msg: str) -> None:
    """progress on stderr (the JSON line on stdout stays the only stdout output)"""
    if int(os.environ.get("RANK", "0")) == 0:
        print(f"[bench {time.time() - _T0:7.1f} s] {msg}", file=sys.stderr, flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--reads", type=int, default=100_000_000)
    ap.add_argument("--k", type=int, default=45, help="CLI k (graph k = k-1, megagta.py:815-816)")
    ap.add_argument("--genes", default="rplB:277,nirK:360")
    ap.add_argument("--cpu-sample", type=int, default=1_000_000)
    ap.add_argument("--seeds", type=int, default=60000, help="seed k-mers per gene of the A* leg (0 = skip the search leg)")
    ap.add_argument("--e2e-reads", type=int, default=2_000_000, help="reads of the reads->contigs leg through megagta.py (0 = skip)")
    ap.add_argument("--e2e-large-reads", type=int, default=10_000_000, help="a larger reads->contigs run, ours only (0 = skip; skipped anyway when the bench run is already late)")
    ap.add_argument("--e2e-ref-reads", type=int, default=200_000, help="small set on which the reference's thread count is chosen before it runs on the e2e set (0 = no reference run)")
    ap.add_argument("--product-seeds", type=int, default=60_000, help="findstart seeds per gene of the product-mode search leg (0 = skip)")
    ap.add_argument("--denovo", action="store_true", help="also run the denovo leg above 20 M reads (half a minute at 100 M)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--check-stream", action="store_true", help="md5 of the edge records every rank holds after the timed steps (the gathered stream with "
                                                                "--gpus N, the resident one with one rank) goes into the line: tests compare the two")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    import torch
    dist = None
    # rehearsal of the N > 1 path on a one-GPU box: MEGAGTA_DIST_BACKEND=gloo MEGAGTA_DEVICE=0 (RCCL refuses two ranks per device)
    backend = os.environ.get("MEGAGTA_DIST_BACKEND", "nccl")
    if "MEGAGTA_DEVICE" in os.environ:
        local_rank = int(os.environ["MEGAGTA_DEVICE"])
    torch.cuda.set_device(local_rank)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    from megagta_amd import api, synth
    from megagta_amd import dist as mdist
    k = args.k - 1
    L = 150
    gene_specs = tuple((g.split(":")[0], int(g.split(":")[1])) for g in args.genes.split(","))
    # identical synthetic read set on every rank (seeded), generated and packed on the device
    t0 = time.time()
    host_sample = max(args.cpu_sample, 1) if rank == 0 and world == 1 else 1      # (the CPU baseline's sample; also the reads the product-mode seeds come from)
    mg = synth.make_metagenome_device(args.reads, L, gene_specs, seed=1, device=f"cuda:{local_rank}", host_sample=host_sample)
    t_gen = time.time() - t0
    note(f"{args.reads} reads generated and packed on the device in {t_gen:.1f} s")

    ctx = api.Context(local_rank)
    rd = ctx.adopt_reads(mg.packed.data_ptr(), mg.n_words, mg.start.data_ptr(), mg.n_reads, keepalive=(mg.packed, mg.start))
    b0, b1 = mdist.bucket_share(rank, world)

    last_whole = [None]

    def step():
        g = ctx.build_sdbg(rd, k, collect=False, bucket_range=(b0, b1))
        if args.check_stream and world == 1:
            last_whole[0] = [api.export_records_to_torch(ctx)]
        if world > 1:
            # the path's one exchange: every rank receives every shard of the edge stream (RCCL all-gather),
            # device to device: the shard never visits the host.  The library writes the shard on its own stream: torch's stream is
            # drained first (the block it hands out may still be read by the previous step's collectives)
            torch.cuda.current_stream().synchronize()
            shard = api.export_records_to_torch(ctx)
            whole = mdist.all_gather_bytes(shard, piece=256 << 20)   # (torch.cat of the pieces = the stream; the graph loader takes them as they are)
            last_whole[0] = whole if args.check_stream else None
            del whole
        return g.stats

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    if world > 1:
        ctx.keep_stream(True)                               # a shard that takes several memory-bound passes is handed over whole
    for _ in range(args.warmup):
        step()
    fence()
    note("warm-up done")
    t = time.time()
    stats = []
    for _ in range(args.steps):
        stats.append(step())
    fence()
    dt = time.time() - t
    note(f"{args.steps} build steps: {dt / args.steps * 1e3:.1f} ms each, {stats[-1]['n_passes']} pass(es)")
    stream_md5 = None
    if args.check_stream and last_whole[0] is not None:      # (outside the timed region) what this rank holds after the exchange, rank by rank in bucket order
        hsh = hashlib.md5()
        for piece in last_whole[0]:
            hsh.update(piece.cpu().numpy().tobytes())
        stream_md5 = hsh.hexdigest()
        last_whole[0] = None
    if world > 1:
        td = torch.tensor([dt], device="cuda", dtype=torch.float64)
        dist.all_reduce(td, op=dist.ReduceOp.MAX)
        dt = float(td.item())

    # ---- A* leg: graph + HMMs replicated (every rank builds the whole graph from its resident reads: the stream of all passes stays on
    # the device), seeds shard by gene, then round-robin; one all-gather of contigs
    search = None
    findstart_leg = None
    denovo_leg = None
    if args.seeds > 0:
        from megagta_amd import hmm as hmmlib
        from megagta_amd import findstart as fsm
        ctx.keep_stream(True)
        tg = time.time()
        gst = ctx.build_sdbg(rd, k, collect=False).stats
        graph = api.Graph(ctx, None, k)                 # row f-4: the stream never leaves the device between build and search
        t_graph = time.time() - tg
        note(f"graph of {graph.size} edges resident ({t_graph:.1f} s incl. the build)")
        ctx.keep_stream(False)
        ctx.release_scratch()                           # the build's key buffers make room for the searches' pool
        td = tempfile.mkdtemp(prefix="mgta_bench_")
        synth.write_gene_models(mg.genes, td)
        hm, seeds, product_seeds = [], [], []
        fs_ms, fs_hits = 0.0, 0
        for gi, gene in enumerate(mg.genes):
            d = os.path.join(td, gene.name)
            hm.append((api.DeviceHmm(ctx, hmmlib.parse_hmm(os.path.join(d, "for_enone.hmm"))), api.DeviceHmm(ctx, hmmlib.parse_hmm(os.path.join(d, "rev_enone.hmm")))))
            seeds.append(synth.synthetic_seeds(gene, args.k, args.seeds, seed=4 + gi))
            # seed finder (row f-2) on the reads already resident for the build: kernel time of one scan per gene
            fwords, _fpos = fsm.reference_words(os.path.join(d, "ref_aligned.faa"), args.k // 3)
            fhits, fms = fsm.find_hits(ctx, rd, True, args.k, fsm.pack_words(fwords, args.k // 3))
            fs_ms += fms
            fs_hits += int(fhits.size)
            # the product's seeds of the reads the host holds a copy of (the CPU-baseline sample): unique k-mers in sorted order, as
            # `megagta findstart` writes them, for the product-mode search leg below
            ps_all = product_seed_list(fhits, mg.sample_reads, fwords, _fpos, args.k) if rank == 0 and world == 1 else []
            # (a contiguous block of the sorted list, at most --product-seeds per gene: neighbours in that order share their paths, which
            # is what the ordered window lives on; the whole list of the sample is a quarter of a million seeds per gene at 100 M reads)
            lo_ = max(0, (len(ps_all) - args.product_seeds) // 2)
            product_seeds.append(ps_all[lo_:lo_ + args.product_seeds])
            del fhits
        findstart_leg = {"ms_kernel": fs_ms, "windows_per_s": len(mg.genes) * args.reads * (L - args.k + 1) * 2 / (fs_ms * 1e-3), "hits": fs_hits,
                         "note": "mgta_findstart, both strands, k=%d, one scan per gene (%d genes)" % (args.k, len(mg.genes))}
        shutil.rmtree(td, ignore_errors=True)
        note(f"seed scans: {fs_ms:.1f} ms, {fs_hits} hits; {sum(len(x) for x in seeds)} synthetic seeds")
        share = mdist.gene_seed_share([len(s) for s in seeds], rank, world)
        last_contigs = {}

        def sstep(genes=None):
            tot = {"n_expansions": 0, "ms_kernel": 0.0, "n_retries": 0, "n_grown": 0, "pool_used": 0, "ms_queue_drained": 0.0, "per_gene": {}}
            mine_all = []
            for gi in (range(len(mg.genes)) if genes is None else genes):
                mine = share[gi]
                kmers, states = [seeds[gi][i][0] for i in mine], [seeds[gi][i][1] - 1 for i in mine]
                cont, offs, st = api.astar_search_packed(graph, hm[gi][0], hm[gi][1], kmers, states, 20, 0.5) if len(mine) else \
                    (np.zeros(0, np.uint8), np.zeros(1, np.int64), None)
                if st:
                    for key in ("n_expansions", "ms_kernel", "n_retries", "n_grown", "ms_queue_drained"):
                        tot[key] += st[key]
                    tot["per_gene"][mg.genes[gi].name] = {"expansions": st["n_expansions"], "ms_kernel": st["ms_kernel"], "ms_queue_drained": st["ms_queue_drained"],
                                                          "longest_search_expansions": st["max_search_expansions"]}
                    tot["pool_used"] = max(tot["pool_used"], st["pool_used"])
                if world > 1:
                    mine_all.append((len(seeds[gi]), mine, cont, offs))
                elif len(mine):
                    last_contigs[gi] = (cont, offs)
            if world > 1:                                   # the path's one exchange: ONE all-gather of the contigs of every gene, at the end
                mdist.all_gather_all_genes([m[0] for m in mine_all], [m[1] for m in mine_all], [m[2] for m in mine_all], [m[3] for m in mine_all])
            return tot

        # the yardstick of the leg's roofline first, while its 8 GB table still fits beside the graph (the searches' pool takes the rest)
        line_probe = None
        if rank == 0:
            line_probe = random_line_probe(ctx)
            note(f"random 128-byte lines: peak {line_probe['peak_gb_per_s']:.0f} GB/s at {line_probe['peak_at_lines_in_flight_per_cu']} lines in flight per CU; "
                 f"one dependent line {line_probe['idle_dependent_ns']:.0f} ns on the idle chip, "
                 + ", ".join(f"{r['ns_per_line']:.0f} ns at {r['chains_per_cu']} chains/CU" for r in line_probe["dependent"][1:]))
        # warm-up: the first gene alone at 100 M reads (it obtains the pool at its full size and loads the kernels; a whole step takes
        # most of a minute there)
        w0 = sstep([0] if args.reads > 20_000_000 else None)
        fence()
        note(f"search warm-up: {w0['n_expansions']} expansions, {w0['ms_kernel']:.0f} ms on the device")
        t = time.time()
        n_s = 1 if args.reads > 20_000_000 else max(1, args.steps // 2)      # (a step of the search leg takes ~25 s at 100 M reads)
        sst = [sstep() for _ in range(n_s)]
        fence()
        sdt = (time.time() - t) / n_s
        note(f"search: {sdt:.2f} s per step")
        nexp = torch.tensor([float(sst[-1]["n_expansions"]), sdt], dtype=torch.float64, device="cuda")
        if world > 1:
            ne = nexp[:1].clone()
            dist.all_reduce(ne)
            tm = nexp[1:].clone()
            dist.all_reduce(tm, op=dist.ReduceOp.MAX)
            nexp = torch.cat([ne, tm])
        rate = float(nexp[0]) / float(nexp[1])
        search = {"value": rate, "unit": "HMM-scored node expansions/s", "n_seeds": sum(len(s) for s in seeds), "genes": [g.name for g in mg.genes],
                  "graph_edges": int(graph.size), "graph_build_and_load_s": t_graph, "graph_passes": gst["n_passes"],
                  "expansions_per_step": float(nexp[0]), "ms_per_step": float(nexp[1]) * 1e3, "cache_mode": "cold (every seed independent)",
                  "lanes_per_search": 8 if max(len(x) for x in seeds) >= 32768 else 16, "searches_in_flight_per_gpu": 16384 if max(len(x) for x in seeds) >= 32768 else 8192,
                  "ms_kernel": sst[-1]["ms_kernel"], "retries": sst[-1]["n_retries"], "searches_grown_in_place": sst[-1]["n_grown"],
                  "warmup_expansions": w0["n_expansions"],
                  # a batch cannot end before its longest search does: from the moment the last seed is TAKEN the launch only finishes what is in flight
                  "tail": {"ms_kernel": sst[-1]["ms_kernel"], "ms_until_the_last_seed_was_taken": sst[-1]["ms_queue_drained"],
                           "tail_fraction_of_kernel_time": 1.0 - sst[-1]["ms_queue_drained"] / max(1e-9, sst[-1]["ms_kernel"]), "per_gene": sst[-1]["per_gene"],
                           "note": "mgta_astar_stats.ms_queue_drained; one search alone takes 14-17 us per expansion (DESIGN.md 5)"},
                  "pool_used_GB": sst[-1]["pool_used"] / 1e9}
        if rank == 0:
            # roofline of the leg.  Bound: random 128-byte lines (graph lines, heap blocks, hash lines, nodes) -- not the 8 TB/s stream rate and
            # not an MFMA peak.  `peak` = the most this device delivers on independent random lines (the library's HIP probe, measured in this
            # run before the searches took their pool); `achieved` = the lines the kernel really misses (TCC_MISS per expansion of the round's
            # PMC pass x 128 B x this run's rate) when that pass is of this workload and kernel source, else SURVEY.md §8d's algorithmic
            # 510 B per expansion (170 B x (1 + d1 + d1 d2), unbranched).  Both fractions are <= 1 by construction: the probe saturates the
            # memory system with the same access shape.  `chase` = what the kernel's OWN concurrency (one dependent chain per search slot)
            # could reach if an expansion were nothing but its line fetches.
            lanes = 8 if max(len(x) for x in seeds) >= 32768 else 16
            slots_per_cu = 8 * (64 // lanes)
            pk = line_probe["peak_gb_per_s"]
            chase = min(line_probe["dependent"], key=lambda r: abs(r["chains_per_cu"] - slots_per_cu))
            rl = {"bound": "hbm (random 128-byte lines)", "bytes_per_expansion_algorithmic": 510, "achieved_algorithmic": rate * 510 / 1e9,
                  "achieved": rate * 510 / 1e9, "achieved_is": "algorithmic bytes", "peak": pk, "unit": "GB/s", "frac": rate * 510 / 1e9 / pk,
                  "peak_note": "mgta_probe_random_lines in this run: independent random 128-byte lines over an %d GB table, groups of 8 lanes per line, best of a sweep "
                               "over 4 ... 2048 lines in flight per CU (reached at %d)" % (line_probe["table_bytes"] >> 30, line_probe["peak_at_lines_in_flight_per_cu"]),
                  "probe": line_probe,
                  "chase": {"search_slots_per_cu": slots_per_cu, "chains_per_cu": chase["chains_per_cu"], "ns_per_dependent_line": chase["ns_per_line"],
                            "gb_per_s": chase["gb_per_s"], "note": "pointer chase at the kernel's own concurrency: one dependent line per search slot at a time"},
                  "hbm_stream_peak": HBM_PEAK_GBS, "frac_of_stream_peak": rate * 510 / 1e9 / HBM_PEAK_GBS}
            cp = os.path.join(ROOT, "profiles", "astar_counters_latest.json")
            if os.path.exists(cp):
                try:
                    cj = json.load(open(cp))
                    same = cj.get("reads") == args.reads and cj.get("graph_k") == k and cj.get("lanes_per_search") == lanes and \
                        cj.get("source_signature") == source_signature(*ASTAR_SOURCES)
                    if same:
                        mpe = cj["TCC_MISS_sum"] / cj["expansions"]
                        moved = rate * mpe * 128 / 1e9
                        rl["traffic"] = {"l2_misses_per_expansion": mpe, "line_bytes": 128, "achieved": moved, "unit": "GB/s",
                                         "over_algorithmic": mpe * 128 / 510, "frac_of_random_line_peak": moved / pk,
                                         "frac_of_chase_at_kernel_concurrency": moved / chase["gb_per_s"],
                                         "note": "TCC_MISS_sum / expansions of the PMC pass `%s` (collected %s at commit %s) x this run's rate: priced on "
                                                 "the lines it actually misses" % (cj.get("command"), cj.get("collected"), cj.get("commit"))}
                        rl["achieved"], rl["achieved_is"], rl["frac"] = moved, "lines missed (PMC) x 128 B", moved / pk
                    else:
                        rl["traffic"] = None
                        rl["traffic_source"] = "profiles/astar_counters_latest.json is of another workload, lane group or kernel source: not quoted"
                except Exception:
                    pass
            search["roofline"] = rl
        if rank == 0 and world == 1 and last_contigs:
            search["membership"] = contig_membership(graph, last_contigs, k)
            note(f"membership of the returned contigs' (k+1)-mers in the graph: {search['membership']}")
        if rank == 0 and world == 1 and sum(len(x) for x in product_seeds) > 0:
            # the mode `megagta search` runs in: findstart's seeds in its order, shared term_nodes caches under the ordered-commit window
            # (window and cost term chosen by the number of seeds, as the binary chooses them), on the same resident graph
            from megagta_amd import search_dist as sdm
            tot_e, tot_ms, per_gene = 0, 0.0, {}
            t = time.time()
            for gi, gene in enumerate(mg.genes):
                ps = product_seeds[gi]
                if not ps:
                    continue
                window, rate = sdm.window_and_rate(len(ps))
                _, _, st = api.astar_search_packed(graph, hm[gi][0], hm[gi][1], [x[0] for x in ps], [x[1] - 1 for x in ps], 20, 0.5, cache_mode=window, cost_rate=rate)
                tot_e += st["n_expansions"]; tot_ms += st["ms_total"]
                per_gene[gene.name] = {"seeds": len(ps), "window": window, "cost_rate": rate, "expansions": st["n_expansions"], "ms": st["ms_total"],
                                       "restarted_in_place": st["n_retries"]}
            pdt = time.time() - t
            search["product_mode"] = {"value": tot_e / max(1e-9, tot_ms * 1e-3), "unit": "HMM-scored node expansions/s", "seeds_per_s": sum(len(x) for x in product_seeds) / pdt,
                                      "seconds": pdt, "genes": per_gene,
                                      "note": "findstart's seeds of the first %d reads (sorted, unique; a contiguous block of at most %d per gene), default mode of "
                                              "`megagta search` (ordered-commit window + cost term) on the %d-edge graph; `value` above is the cold mode on "
                                              "synthetic seeds" % (mg.sample_reads.shape[0], args.product_seeds, graph.size)}
            note(f"search, product mode: {sum(len(x) for x in product_seeds)} seeds in {pdt:.1f} s, {tot_e / max(1e-9, tot_ms * 1e-3) / 1e6:.1f} M expansions/s")
        if world == 1 and (args.reads <= 20_000_000 or args.denovo):
            # row f-1: tips, bubbles, unitigs on the same resident graph (last: it consumes the validity bits).  The searches' pool goes
            # first: the bubble rounds size their windows by the free memory
            ctx.release_scratch()
            _, dst = graph.denovo(150, False, k + 2)
            denovo_leg = {"edges": int(graph.size), "ms_tips": dst["ms_tips"], "ms_bubbles": dst["ms_bubbles"], "ms_unitigs": dst["ms_unitigs"],
                          "tips": dst["n_tips"], "bubbles": dst["n_bubbles"], "bubble_rounds": dst["n_bubble_rounds"], "contigs": dst["n_contigs"],
                          "edges_per_s": graph.size / max(1e-9, (dst["ms_tips"] + dst["ms_bubbles"] + dst["ms_unitigs"]) * 1e-3),
                          "note": "mgta_denovo --max_tip_len 150 on the build leg's graph (one-thread-reference result, computed on the device)"}
        graph.free()

    if rank == 0:
        s = stats[-1]
        n_kmers = s["n_kmers"]                       # every rank scans all reads: whole-job k-mers per step
        ms_step = dt / args.steps * 1e3
        value = n_kmers / (ms_step * 1e-3) / 1e9
        # the two big kernels of the build: radix_scatter (P launches per pass: the most significant bytes) and local_sort (one launch per
        # pass: every remaining digit in LDS).  Each launch reads and writes every key of the pass once = 16W bytes per
        # (k+1)-mer occurrence.  Durations: HIP events recorded by the library on its own stream.
        W = s["words_per_key"]
        items_per_launch = s["n_items"] / max(1, s["n_passes"])
        alg_bytes = items_per_launch * 4 * W * 2
        launches = sum(x["n_sort_launches"] for x in stats)
        ms_scatter = sum(x["ms_sort_scatter"] for x in stats) / max(1, launches)
        ms_local = sum(x["ms_local_sort"] for x in stats) / max(1, sum(x["n_passes"] for x in stats))
        scatter_total = sum(x["ms_sort_scatter"] for x in stats)
        local_total = sum(x["ms_local_sort"] for x in stats)
        # since round 5 every scatter launch but the last of a pass also writes the NEXT pass's digit of every key (1 B/key, W <= 7) so that the
        # census in between reads 1 byte instead of the key; `achieved` stays on SURVEY 8(d)'s bytes (2 x 4W per key), the bytes including
        # that side array are reported next to it
        n_pass_total = sum(x["n_passes"] for x in stats)
        side_on = os.environ.get("MGTA_SORT_SIDE", "1") != "0" and W <= 7 and launches > n_pass_total
        side_bytes = items_per_launch * (launches - n_pass_total) / max(1, launches) if side_on else 0.0
        dom_name, dom_ms = ("local_sort_kernel", ms_local) if local_total >= scatter_total else ("radix_scatter_kernel", ms_scatter)
        achieved = alg_bytes / (dom_ms * 1e-3) / 1e9 if dom_ms > 0 else 0.0
        # HBM bytes per launch of the dominant kernel from the PMC passes (they cannot be collected inline: separate rocprofv3 runs,
        # scripts/profile_r04.sh) -- quoted only when that profile is of THIS workload and of THESE kernel sources, else null with the reason
        traffic, traffic_note = None, "no PMC summary (profiles/traffic_latest.json)"
        tp = os.path.join(ROOT, "profiles", "traffic_latest.json")
        if os.path.exists(tp):
            tj = json.load(open(tp))
            if tj.get("reads") != args.reads or tj.get("graph_k", k) != k:
                traffic_note = f"profiles/traffic_latest.json is of another workload ({tj.get('reads')} reads)"
            elif tj.get("source_signature") != source_signature(*BUILD_SOURCES):
                traffic_note = "profiles/traffic_latest.json is STALE: the build kernels changed since it was collected (%s)" % tj.get("collected", "no date")
            else:
                traffic = tj.get(dom_name + "_bytes_per_launch")
                traffic_note = "2 x FETCH_SIZE + WRITE_SIZE per launch, %s, collected %s at commit %s" % (tj.get("_source", ""), tj.get("collected"), tj.get("commit"))
        edges_per_kmer = s["n_edges"] / max(1, n_kmers) * world
        gene_txt = " + ".join(g[0] for g in gene_specs)
        out = {
            "metric": f"HMM-scored node expansions/sec + SdBG-build Gk-mer/s, k={args.k}, {args.reads // 1_000_000}Mx{L}bp",
            "value": value, "unit": "Gk-mer/s (SdBG build)", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_step, "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "u32",
            "data": "synthetic",
            "config": {"workload": f"{gene_txt}, {args.reads} x {L}bp synthetic reads, CLI k={args.k} (graph k={k}), -c 1, "
                                   f"{'bucket-range sharded, all-gather of record shards' if world > 1 else '1x MI355X'}",
                       "reads": args.reads, "read_len": L, "graph_k": k, "n_kmers": n_kmers, "n_items": s["n_items"],
                       "n_edges_rank0": s["n_edges"], "passes": s["n_passes"]},
            "roofline": {"bound": "hbm", "kernel": dom_name, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_note, "avg_launch_ms": dom_ms,
                         "algorithmic_bytes_per_launch": alg_bytes,
                         "other_kernels": {"radix_scatter_kernel": {"avg_launch_ms": ms_scatter, "launches_per_step": launches / args.steps,
                                                                    "achieved": alg_bytes / (ms_scatter * 1e-3) / 1e9 if ms_scatter > 0 else 0.0,
                                                                    "achieved_incl_side_digits": (alg_bytes + side_bytes) / (ms_scatter * 1e-3) / 1e9 if ms_scatter > 0 else 0.0,
                                                                    "side_digit_bytes_per_launch": side_bytes},
                                           "local_sort_kernel": {"avg_launch_ms": ms_local, "launches_per_step": s["n_passes"],
                                                                 "achieved": alg_bytes / (ms_local * 1e-3) / 1e9 if ms_local > 0 else 0.0}}},
            "whole_build": {"algorithmic_bytes_per_kmer": b_build(k, L, edges_per_kmer),
                            "achieved_GBps": n_kmers * b_build(k, L, edges_per_kmer) / (ms_step * 1e-3) / 1e9,
                            "frac_of_hbm_peak": n_kmers * b_build(k, L, edges_per_kmer) / (ms_step * 1e-3) / 1e9 / HBM_PEAK_GBS,
                            "phase_ms": {p: s[p] for p in ("ms_count", "ms_gen", "ms_sort", "ms_emit", "ms_total")},
                            "oversized_segments": s["n_big_segments"],
                            # what does not shrink with the number of GPUs: a bucket-sharded build scans ALL reads on every rank (every read has
                            # items in every bucket range) -- the count and the key generation of a step are the floor of the 1 -> N curve
                            "rank_invariant_floor_ms": s["ms_count"] + s["ms_gen"],
                            "rank_invariant_floor_note": "count + key generation: every rank of a bucket-sharded build scans all reads; "
                                                         "sort + emit shrink with N, this part does not (SURVEY.md 8e)",
                            # ... and in a `megagta.py --gpus N` run: `denovo` of the intermediate k (GPU 0 only; 26 + 23 s at 100 M reads,
                            # profiles/r05/e2e_100M_reads_to_seeds_steps.log) and the host steps do not shrink either; `findstart` is
                            # 0.17 s of kernel per gene at this size (`findstart` leg of this line): nothing to shard
                            "driver_run_non_scaling_note": "megagta.py --gpus N shards buildgraph (buckets) and search (genes, then seeds); denovo of the intermediate k "
                                                           "runs on GPU 0: 49 s of the 81 s reads -> seeds at 100 M reads (profiles/r05); findstart's kernel is 0.17 s per gene",
                            "pcie_inclusive_note": "inputs resident; uploading the packed reads (0.25 B/base + 8 B/read at ~55 GB/s) and returning "
                                                   "2 B/edge would add ~%.0f ms per build" % ((args.reads * (L * 0.25 + 8) + s["n_edges"] * 2) / 55e9 * 1e3)},
            "input_generation_s": t_gen,
            "host": host_cores(),
        }
        if stream_md5 is not None:
            out["stream_md5"] = stream_md5
        if search is not None:
            out["search"] = search
            out["findstart"] = findstart_leg
            if denovo_leg is not None:
                out["denovo"] = denovo_leg
        if world == 1:
            # free the device before the child processes of the e2e leg ask for it
            rd.free()
            rd._keep = None
            ctx.release_scratch()
            mg.packed = mg.start = None
            torch.cuda.empty_cache()
            if args.e2e_reads > 0:
                try:
                    # the driver clears the ~250 GB this process has just released before it hands them to the next process: the first
                    # allocations of the leg's child processes waited 5 s for that (profiles/r02/vmm_probe.log: 20-90 ms/GB)
                    time.sleep(8 if args.reads > 20_000_000 else 1)
                    note("e2e leg ...")
                    out["e2e"] = e2e_leg(gene_specs, args.e2e_reads, args.e2e_ref_reads, f"cuda:{local_rank}", n_large=args.e2e_large_reads,
                                         large_deadline=_T0 + 350.0, hard_stop=_T0 + 520.0)     # (the leg takes ~100 s and the CPU baselines ~45 s after it: the driver's call ends at 600 s)
                    note("e2e leg done")
                except Exception as e:                                   # the bench line must not die with a leg
                    out["e2e"] = {"error": str(e)[-600:]}
            if not args.no_cpu_baseline:
                note("cpu baseline ...")
                cb = cpu_baseline(mg.sample_reads, k, args.cpu_sample, ctx, mg.genes)
                if "_parity" in cb:
                    out["parity_1M"] = cb.pop("_parity")
                    note(f"parity vs the reference's graph of the sample: {out['parity_1M']}")
                if "_search" in cb:
                    sb = cb.pop("_search")
                    if "search" in out:
                        out["search"]["cpu_baseline"] = sb
                    else:
                        out["search_cpu_baseline"] = sb
                    note(f"search cpu baseline: {sb}")
                out["cpu_baseline"] = cb
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
