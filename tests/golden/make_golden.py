#!/usr/bin/env python3
"""Regenerates tests/golden/* from the COMPILED REFERENCE (oracle/_ref/megagta + oracle/_ref/probe).

Run in the build container only (needs /root/reference to have been built by `make -C oracle ref`):

    python tests/golden/make_golden.py

Inputs are seeded synthetic data (megagta_amd.synth); outputs are small data fixtures: inputs
(reads.lib.bin, HMM text, seed files) and the reference's answers (edge-stream digests + leading
records, parsed HMM tables, graph navigation answers, per-seed A* results, raw contig FASTA).
No reference source text is stored, only data.
"""
from __future__ import annotations

import gzip
import hashlib
import json
import os
import shutil
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from megagta_amd import synth          # noqa: E402
from oracle import oracle as O         # noqa: E402  (decoder of .sdbg files only; validated against md5 of raw files below)

REF = os.path.join(ROOT, "oracle", "_ref")
GOLD = os.path.join(ROOT, "tests", "golden")


def run(cmd, **kw):
    return subprocess.run(cmd, check=True, stdout=subprocess.PIPE, stderr=subprocess.PIPE, **kw)


def gz_write(path, data: bytes):
    with gzip.GzipFile(path, "wb", mtime=0) as f:
        f.write(data)


def buildgraph(lib, prefix, k, extra=(), m=1):
    os.makedirs(os.path.dirname(prefix), exist_ok=True)
    run([f"{REF}/megagta", "buildgraph", "-k", str(k), "-m", str(m), "--host_mem", "4000000000", "--mem_flag", "1",
         "--gpu_mem", "0", "--output_prefix", prefix, "--num_cpu_threads", "4", "--num_output_threads", "1",
         "--read_lib_file", lib, *extra])


def stream_fixture(prefix):
    e = O.Stream.read(prefix).edges()
    return dict(k=e.k, words_per_tip=e.words_per_tip, num_edges=int(e.records.size), num_tips=int(e.tips.size // max(1, e.words_per_tip)),
                num_large=int(e.large.size), md5=e.md5(),
                bucket_md5=hashlib.md5(e.bucket_items.astype("<i8").tobytes()).hexdigest(),
                head_records=[int(x) for x in e.records[:256]], large=[int(x) for x in e.large[:64]],
                head_tips=[int(x) for x in e.tips[: 8 * e.words_per_tip]])


def main():
    tmp = tempfile.mkdtemp(prefix="mgta_gold_")
    # ------------------------------------------------------------------ toy: fixed-length reads
    toy = os.path.join(GOLD, "toy")
    shutil.rmtree(toy, ignore_errors=True)
    os.makedirs(toy)
    mg = synth.make_metagenome(6000, 150, (("rplB", 100),), seed=11, reads_per_genome=1000)
    synth.write_fasta(mg.reads, f"{tmp}/reads.fa")
    open(f"{tmp}/reads.lib", "w").write(f"reads.fa\nse {tmp}/reads.fa\n")
    run([f"{REF}/megagta", "buildlib", f"{tmp}/reads.lib", f"{tmp}/reads.lib"])
    shutil.copy(f"{tmp}/reads.lib.bin", f"{toy}/reads.lib.bin")
    open(f"{toy}/reads.lib.lib_info", "w").write(open(f"{tmp}/reads.lib.lib_info").read())
    gl = synth.write_gene_models(mg.genes, f"{tmp}/genes")
    for n in ("for_enone.hmm", "rev_enone.hmm", "ref_aligned.faa"):
        shutil.copy(f"{tmp}/genes/rplB/{n}", f"{toy}/{n}")
    streams = {}
    for k in (29, 35, 44):
        buildgraph(f"{tmp}/reads.lib", f"{tmp}/k{k}/{k}", k)
        streams[str(k)] = stream_fixture(f"{tmp}/k{k}/{k}")
    json.dump(streams, open(f"{toy}/sdbg_streams.json", "w"), indent=0)
    # stage 1 (solid-edge counting, -m >= 2) with and without mercy edges (cx1_read2sdbg_s1.cpp, s2.cpp:106-250)
    solid = {}
    for k, m, mercy in ((29, 2, False), (29, 2, True), (44, 2, False), (44, 2, True), (44, 3, True), (35, 4, True),
                        (111, 2, True), (120, 2, False), (127, 2, True)):      # 10- / 11-word sort records, up to kMaxK (definitions.h:56)
        tag = f"k{k}_m{m}_{'mercy' if mercy else 'nomercy'}"
        buildgraph(f"{tmp}/reads.lib", f"{tmp}/s1_{tag}/{k}", k, extra=(("--need_mercy",) if mercy else ()), m=m)
        fx = stream_fixture(f"{tmp}/s1_{tag}/{k}")
        fx["counting_md5"] = hashlib.md5(open(f"{tmp}/s1_{tag}/{k}.counting", "rb").read()).hexdigest()
        fx["counting_head"] = open(f"{tmp}/s1_{tag}/{k}.counting").read().splitlines()[:40]
        solid[tag] = fx
    json.dump(solid, open(f"{toy}/sdbg_streams_solid.json", "w"), indent=0)
    # parsed HMM tables + heuristic
    for tag in ("for", "rev"):
        gz_write(f"{toy}/hmm_{tag}.txt.gz", run([f"{REF}/probe", "hmm", f"{toy}/{tag}_enone.hmm"]).stdout)
    gz_write(f"{toy}/codon.txt.gz", run([f"{REF}/probe", "codon"]).stdout)
    # graph navigation answers
    gz_write(f"{toy}/graph_k44.txt.gz", run([f"{REF}/probe", "graph", f"{tmp}/k44/44", "1500", "5"]).stdout)
    # seeds from the reference's findstart; (k+1)-mers for IndexBinarySearchEdge incl. absent ones
    seeds = run([f"{REF}/megagta", "findstart", f"{toy}/ref_aligned.faa", f"{tmp}/reads.lib.bin", "45", "1"]).stdout
    open(f"{toy}/44_rplB_starting_kmers.txt", "wb").write(seeds)
    kmers = [l.split(b"\t")[3].decode() for l in seeds.splitlines()]
    rng = np.random.default_rng(3)
    for _ in range(64):
        kmers.append("".join("ACGT"[i] for i in rng.integers(0, 4, 45)))
    open(f"{tmp}/kmers.txt", "w").write("\n".join(kmers) + "\n")
    gz_write(f"{toy}/index_k44.txt.gz", run([f"{REF}/probe", "index", f"{tmp}/k44/44", f"{tmp}/kmers.txt"]).stdout)
    # A*: per-seed cold-cache and sequential warm-cache results, plus the reference's own FASTA (1 thread)
    shutil.copy(f"{toy}/44_rplB_starting_kmers.txt", f"{tmp}/k44/44_rplB_starting_kmers.txt")
    for mode in ("cold", "warm"):
        out = run([f"{REF}/probe", "astar", f"{tmp}/k44/44", f"{toy}/for_enone.hmm", f"{toy}/rev_enone.hmm",
                   f"{toy}/44_rplB_starting_kmers.txt", "20", "0.5", mode]).stdout
        gz_write(f"{toy}/astar_{mode}.txt.gz", out)
    open(f"{tmp}/gene_list.txt", "w").write(f"rplB {toy}/for_enone.hmm {toy}/rev_enone.hmm {toy}/ref_aligned.faa\n")
    run([f"{REF}/megagta", "search", f"{tmp}/k44/44", f"{tmp}/gene_list.txt", f"{tmp}/k44/44", f"{tmp}/k44/44", "20", "0.5", "1"])
    gz_write(f"{toy}/44_raw_contigs_rplB.fasta.gz", open(f"{tmp}/k44/44_raw_contigs_rplB.fasta", "rb").read())
    # prune_len 0 variant (no heuristic pruning branch, hmm_graph_search.h:313-325) on a few seeds
    open(f"{tmp}/seeds_few.txt", "wb").write(b"\n".join(seeds.splitlines()[:12]) + b"\n")
    out = run([f"{REF}/probe", "astar", f"{tmp}/k44/44", f"{toy}/for_enone.hmm", f"{toy}/rev_enone.hmm",
               f"{tmp}/seeds_few.txt", "0", "0.5", "cold"]).stdout
    gz_write(f"{toy}/astar_cold_prune0.txt.gz", out)

    # ------------------------------------------------------------------ ragged: edge cases of the build
    rag = os.path.join(GOLD, "ragged")
    shutil.rmtree(rag, ignore_errors=True)
    os.makedirs(rag)
    rng = np.random.default_rng(5)
    genome = rng.integers(0, 4, 3000)
    reads = []
    for _ in range(700):                                   # ragged lengths 20..150 (some shorter than k+1)
        L = int(rng.integers(20, 151))
        p = int(rng.integers(0, 3000 - L))
        r = genome[p:p + L].copy()
        if rng.random() < 0.5:
            r = 3 - r[::-1]
        e = rng.random(L) < 0.01
        r[e] = (r[e] + rng.integers(1, 4, int(e.sum()))) & 3
        reads.append("".join("ACGT"[i] for i in r))
    hot = "".join("ACGT"[i] for i in genome[100:230])
    reads += [hot] * 300                                   # multiplicity > 254 -> large-multiplicity records
    half = genome[500:515]
    pal = "".join("ACGT"[i] for i in np.concatenate([half, 3 - half[::-1]]))   # 30-mer == its reverse complement
    reads += ["".join("ACGT"[i] for i in genome[480:500]) + pal + "".join("ACGT"[i] for i in genome[530:560])] * 3
    reads += [pal, "ACGTNNACGTNNACGTACGTTTGACCAGTANNNNCATGACCGATAGGACCATGACATAGGAC", "A" * 45, "ACGT" * 12 + "A"]
    with open(f"{tmp}/ragged.fa", "w") as f:
        for i, r in enumerate(reads):
            f.write(f">x{i}\n{r}\n")
    open(f"{tmp}/ragged.lib", "w").write(f"ragged.fa\nse {tmp}/ragged.fa\n")
    run([f"{REF}/megagta", "buildlib", f"{tmp}/ragged.lib", f"{tmp}/ragged.lib"])
    shutil.copy(f"{tmp}/ragged.lib.bin", f"{rag}/reads.lib.bin")
    open(f"{rag}/reads.lib.lib_info", "w").write(open(f"{tmp}/ragged.lib.lib_info").read())
    streams = {}
    for k in (21, 29, 31, 44, 47, 63):
        buildgraph(f"{tmp}/ragged.lib", f"{tmp}/rk{k}/{k}", k)
        streams[str(k)] = stream_fixture(f"{tmp}/rk{k}/{k}")
    json.dump(streams, open(f"{rag}/sdbg_streams.json", "w"), indent=0)
    solid = {}
    for k, m, mercy in ((21, 2, True), (29, 2, True), (31, 3, False), (47, 2, False), (47, 2, True)):
        tag = f"k{k}_m{m}_{'mercy' if mercy else 'nomercy'}"
        try:
            buildgraph(f"{tmp}/ragged.lib", f"{tmp}/rs1_{tag}/{k}", k, extra=(("--need_mercy",) if mercy else ()), m=m)
        except subprocess.CalledProcessError as e:      # the reference itself crashes on some ragged inputs with mercy edges
            solid[tag] = {"reference_crashed": True, "returncode": e.returncode}
            continue
        fx = stream_fixture(f"{tmp}/rs1_{tag}/{k}")
        fx["counting_md5"] = hashlib.md5(open(f"{tmp}/rs1_{tag}/{k}.counting", "rb").read()).hexdigest()
        solid[tag] = fx
    json.dump(solid, open(f"{rag}/sdbg_streams_solid.json", "w"), indent=0)
    # the reference loader's own view of two of these graphs (bit-vector digests + navigation answers):
    # pins the .sdbg/.sdbg_info decoder independently of the oracle's reader
    for k in (29, 47):
        gz_write(f"{rag}/graph_k{k}.txt.gz", run([f"{REF}/probe", "graph", f"{tmp}/rk{k}/{k}", "400", "9"]).stdout)
    shutil.rmtree(tmp)
    print("golden fixtures written under", GOLD)


if __name__ == "__main__":
    main()
