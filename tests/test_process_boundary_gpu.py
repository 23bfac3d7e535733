"""The process-level drop-in boundary: `megagta buildgraph` / `megagta search` (C++ host + libmegagta_hip.so)
driven by the Python-3 driver, next to the stock reference binary for the steps outside the path."""
import os
import subprocess
import sys

import pytest

from megagta_amd import synth
from tests import helpers as H

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.path.join(ROOT, "oracle", "_ref", "megagta")
BIN = os.path.join(ROOT, "megagta_amd", "bin", "megagta")
DRIVER = os.path.join(ROOT, "megagta_amd", "megagta.py")


def _need():
    if not os.path.exists(REF):
        pytest.skip("oracle/_ref/megagta (the prebuilt reference) is not present")
    assert os.path.exists(BIN), "megagta_amd/bin/megagta missing: run __graft_entry__.build()"


@pytest.fixture(scope="module")
def toy_inputs(tmp_path_factory, golden_dir):
    d = tmp_path_factory.mktemp("e2e")
    mg = synth.make_metagenome(6000, 150, (("rplB", 100),), seed=11, reads_per_genome=1000)    # == tests/golden/toy
    synth.write_fasta(mg.reads, str(d / "reads.fa"))
    toy = os.path.join(golden_dir, "toy")
    (d / "gene_list.txt").write_text(f"rplB {toy}/for_enone.hmm {toy}/rev_enone.hmm {toy}/ref_aligned.faa\n")
    return d


def test_single_k_pipeline_matches_reference_artifacts(toy_inputs, oracle, golden_dir):
    _need()
    d = toy_inputs
    out = d / "out1"
    # a single-k run needs no reference binary at all: buildlib, buildgraph, findstart, search, filterbylen, translate are ours
    r = subprocess.run([sys.executable, DRIVER, "-r", str(d / "reads.fa"), "-g", str(d / "gene_list.txt"), "-k", "45", "-o", str(out),
                        "-t", "4", "--min-contig-len", "150"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr + open(out / "log").read()[-2000:]
    toy = os.path.join(golden_dir, "toy")
    # graph files written by OUR buildgraph decode (with the oracle's reader = the reference format) to the reference's stream
    s = oracle.Stream.read(str(out / "k44" / "44")).edges()
    assert s.md5() == H.load_streams(os.path.join(toy, "sdbg_streams.json"))["44"]["md5"]
    # seeds come from OUR findstart (device scan) on the reads.lib.bin the reference's buildlib wrote: the golden seed lines, in
    # sorted order (the reference shuffles its own, fast_kmer_filter.cpp:183)
    seeds = (out / "k44" / "44_rplB_starting_kmers.txt").read_text().splitlines()
    assert seeds == sorted(open(os.path.join(toy, "44_rplB_starting_kmers.txt")).read().splitlines())
    # contigs: one record per seed, names as hmm_graph_search.h:79, sequences == per-seed cold-cache results of the reference
    gold = {g["kmer"].lower(): g["contig"] for g in H.parse_probe_astar(H.gz_lines(os.path.join(toy, "astar_cold.txt.gz")))}
    lines = (out / "k44" / "44_raw_contigs_rplB.fasta").read_text().splitlines()
    assert lines[0::2] == [f">rplB_contig_{2 * i}_contig_{2 * i + 1}" for i in range(len(seeds))]
    assert lines[1::2] == [gold[l.split("\t")[3].lower()] for l in seeds]
    # the last step's text filters against the reference's (filter_by_len.cpp, translate.cpp), header quirks included
    raw = out / "k44" / "44_raw_contigs_rplB.fasta"
    nucl = subprocess.run([REF, "filterbylen", "150"], stdin=open(raw), capture_output=True, text=True, check=True).stdout
    assert (out / "contigs" / "rplB" / "nucl_merged.fasta").read_text() == nucl and nucl.count(">") > 10
    prot = subprocess.run([REF, "translate", str(out / "contigs" / "rplB" / "nucl_merged.fasta")], capture_output=True, text=True, check=True).stdout
    assert (out / "contigs" / "rplB" / "prot_merged.fasta").read_text() == prot
    # driver artefacts
    assert (out / "opts.txt").exists() and (out / "contigs" / "rplB" / "nucl_merged.fasta").exists()
    done = [l.split() for l in (out / "tmp" / "cp.txt").read_text().splitlines()]
    assert [int(a[0]) for a in done] == list(range(len(done))) and all(a[1] == "done" for a in done)


def test_assist_seq_graph_matches_reference(toy_inputs, oracle):
    """multi-k step: k=29 graph (ours) -> denovo (reference) -> k=44 graph with --assist_seq: ours == reference's, bit for bit"""
    _need()
    d = toy_inputs
    w = d / "mk"
    w.mkdir()
    (w / "reads.lib").write_text(f"reads.fa\nse {d / 'reads.fa'}\n")
    run = lambda cmd: subprocess.run(cmd, check=True, capture_output=True)
    run([REF, "buildlib", str(w / "reads.lib"), str(w / "reads.lib")])
    common = ["-m", "1", "--host_mem", "4000000000", "--mem_flag", "1", "--gpu_mem", "0", "--num_cpu_threads", "4",
              "--num_output_threads", "1", "--read_lib_file", str(w / "reads.lib")]
    run([BIN, "buildgraph", "-k", "29", "--output_prefix", str(w / "29")] + common)
    run([REF, "denovo", "-s", str(w / "29"), "-o", str(w / "29"), "-t", "4", "--min_standalone", "400", "--max_tip_len", "150",
         "--min_contig", "45"])
    assert os.path.getsize(w / "29.contigs.fa") > 0
    run([BIN, "buildgraph", "-k", "44", "--output_prefix", str(w / "ours44"), "--assist_seq", str(w / "29.contigs.fa")] + common)
    run([REF, "buildgraph", "-k", "44", "--output_prefix", str(w / "ref44"), "--assist_seq", str(w / "29.contigs.fa")] + common)
    a, b = oracle.Stream.read(str(w / "ours44")).edges(), oracle.Stream.read(str(w / "ref44")).edges()
    assert a.md5() == b.md5() and a.records.size > 0
    # and the reference's own `search` accepts our graph files
    (w / "gl.txt").write_text((d / "gene_list.txt").read_text())
    with open(w / "ours44_rplB_starting_kmers.txt", "w") as f:
        subprocess.run([REF, "findstart", (d / "gene_list.txt").read_text().split()[3], str(w / "reads.lib.bin"), "45", "1"], stdout=f,
                       check=True, stderr=subprocess.DEVNULL)
    run([REF, "search", str(w / "ours44"), str(w / "gl.txt"), str(w / "ours44"), str(w / "refsearch"), "20", "0.5", "1"])
    assert os.path.getsize(w / "refsearch_raw_contigs_rplB.fasta") > 0


def test_error_behaviour():
    _need()
    r = subprocess.run([BIN, "buildgraph", "-k", "44", "-m", "1", "--host_mem", "1e9"], capture_output=True, text=True)
    assert r.returncode == 1 and "No input file!" in r.stderr                         # build_graph.cpp:52-54
    r = subprocess.run([BIN, "buildgraph", "-k", "44", "-m", "1", "--host_mem", "1e9", "--read_lib_file", "x", "--num_cpu_threads", "1"],
                       capture_output=True, text=True)
    assert r.returncode == 1 and "at least 2" in r.stderr                              # :69-71
    r = subprocess.run([BIN, "search", "a", "b"], capture_output=True, text=True)
    assert r.returncode == 1 and "Usage" in r.stderr                                   # search.cpp:72-75
    r = subprocess.run([BIN, "denovo"], capture_output=True, text=True)
    assert r.returncode == 1


def test_buildgraph_min_count_2_with_mercy_matches_reference_binary(toy_inputs, oracle):
    """`megagta buildgraph -m 2 --need_mercy` (what the driver issues for `-c 2`): our .sdbg files and .counting == the reference's"""
    _need()
    d = toy_inputs
    w = d / "m2"
    w.mkdir()
    (w / "reads.lib").write_text(f"reads.fa\nse {d / 'reads.fa'}\n")
    run = lambda cmd: subprocess.run(cmd, check=True, capture_output=True)
    run([REF, "buildlib", str(w / "reads.lib"), str(w / "reads.lib")])
    common = ["-k", "44", "-m", "2", "--host_mem", "4000000000", "--mem_flag", "1", "--gpu_mem", "0", "--num_cpu_threads", "4",
              "--num_output_threads", "1", "--read_lib_file", str(w / "reads.lib"), "--need_mercy"]
    run([BIN, "buildgraph", "--output_prefix", str(w / "ours")] + common)
    run([REF, "buildgraph", "--output_prefix", str(w / "ref")] + common)
    a, b = oracle.Stream.read(str(w / "ours")).edges(), oracle.Stream.read(str(w / "ref")).edges()
    assert a.md5() == b.md5() and a.records.size > 0
    assert (w / "ours.counting").read_text() == (w / "ref.counting").read_text()


@pytest.mark.parametrize("k,with_contigs", [(45, False), (30, True), (72, True)])
def test_findstart_binary_matches_reference_output(golden_dir, k, with_contigs):
    """`megagta findstart <ref> <reads.lib.bin> <k> [threads] [contigs.fa]` (fast_kmer_filter.cpp:49-190): same lines as the reference
    (sorted: the reference shuffles), same usage / missing-file behaviour"""
    import gzip
    assert os.path.exists(BIN), "megagta_amd/bin/megagta missing: run __graft_entry__.build()"
    d = os.path.join(golden_dir, "findstart")
    cmd = [BIN, "findstart", os.path.join(d, "ref_quirks.faa"), os.path.join(d, "reads.lib.bin"), str(k), "2"]
    if with_contigs:
        cmd.append(os.path.join(d, "contigs.fa"))
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    want = gzip.open(os.path.join(d, f"seeds_k{k}{'_contigs' if with_contigs else ''}.txt.gz"), "rt").read().splitlines()
    assert r.stdout.splitlines() == sorted(want)
    r = subprocess.run([BIN, "findstart", "/nonexistent.faa", os.path.join(d, "reads.lib.bin"), "45"], capture_output=True, text=True)
    assert r.returncode == 1 and "doesn't exist" in r.stderr
    r = subprocess.run([BIN, "findstart", os.path.join(d, "ref_quirks.faa"), os.path.join(d, "reads.lib.bin"), "44"], capture_output=True, text=True)
    assert r.returncode != 0 and "multiple of 3" in r.stderr


def test_multi_k_driver_run_stagewise_vs_reference(toy_inputs, oracle):
    """`megagta.py -k 30,36,45`, every step ours.  Each stage is checked against the reference binary fed with the same inputs: the three
    graphs (two of them built with the previous k's contigs as assist sequences), the contigs of the two intermediate k (`denovo`: the
    reference run with one thread, byte for byte), the seeds found in reads plus contigs, and the last step's filters."""
    _need()
    d = toy_inputs
    out = d / "out_mk"
    r = subprocess.run([sys.executable, DRIVER, "-r", str(d / "reads.fa"), "-g", str(d / "gene_list.txt"), "-k", "30,36,45", "-o", str(out),
                        "-t", "4", "--min-contig-len", "150"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr + open(out / "log").read()[-2000:]
    run = lambda cmd, **kw: subprocess.run(cmd, check=True, capture_output=True, **kw)
    lib = str(out / "tmp" / "reads.lib")
    common = ["-m", "1", "--host_mem", "4000000000", "--mem_flag", "1", "--gpu_mem", "0", "--num_cpu_threads", "4", "--num_output_threads", "1",
              "--read_lib_file", lib]
    prev = None
    for k in (29, 35, 44):
        cmd = [REF, "buildgraph", "-k", str(k), "--output_prefix", str(d / f"ref_mk_{k}")] + common
        if prev is not None:
            cmd += ["--assist_seq", str(out / f"k{prev}" / f"{prev}.contigs.fa")]
        run(cmd)
        ours, ref = oracle.Stream.read(str(out / f"k{k}" / f"{k}")).edges(), oracle.Stream.read(str(d / f"ref_mk_{k}")).edges()
        assert ours.md5() == ref.md5() and ours.records.size > 100000, k
        prev = k
    for k, nxt in ((29, 35), (35, 44)):
        run([REF, "denovo", "-s", str(out / f"k{k}" / f"{k}"), "-o", str(d / f"ref_mk_{k}"), "-t", "1", "--min_standalone", "400", "--max_tip_len", "150",
             "--min_contig", str(nxt + 1)])
        assert (out / f"k{k}" / f"{k}.contigs.fa").read_text() == (d / f"ref_mk_{k}.contigs.fa").read_text(), k
        assert (out / f"k{k}" / f"{k}.contigs.fa.info").read_text() == (d / f"ref_mk_{k}.contigs.fa.info").read_text()
        assert os.path.getsize(out / f"k{k}" / f"{k}.contigs.fa") > 1000
    faa = (d / "gene_list.txt").read_text().split()[3]
    ref_seeds = run([REF, "findstart", faa, lib + ".bin", "45", "2", str(out / "k35" / "35.contigs.fa")]).stdout.decode().splitlines()
    ours_seeds = (out / "k44" / "44_rplB_starting_kmers.txt").read_text().splitlines()
    assert ours_seeds == sorted(ref_seeds) and len(ours_seeds) >= 93
    nucl = run([REF, "filterbylen", "150"], stdin=open(out / "k44" / "44_raw_contigs_rplB.fasta")).stdout.decode()
    assert (out / "contigs" / "rplB" / "nucl_merged.fasta").read_text() == nucl and nucl.count(">") > 10


@pytest.fixture(scope="module")
def two_gene_inputs(tmp_path_factory):
    """BASELINE.json configs[2] in small: rplB (M = 277) + nirK (M = 360), both genes in every genome"""
    d = tmp_path_factory.mktemp("cfg3")
    mg = synth.make_metagenome(12000, 150, (("rplB", 277), ("nirK", 360)), seed=23, reads_per_genome=1000, genome_len=12000)
    synth.write_fasta(mg.reads, str(d / "reads.fa"))
    synth.write_gene_models(mg.genes, str(d / "models"))
    return d


def _fasta_seqs(path):
    return [l for l in open(path).read().splitlines() if l and l[0] != ">"]


def test_config3_two_genes_multi_k_stagewise_and_contig_multiset(two_gene_inputs, oracle):
    """`megagta.py -k 30,36,45` with a two-gene gene_list (the configuration BASELINE.json's metric names, in small).  Every stage against
    the reference binary fed the same inputs (as the one-gene test above does); then the raw contigs of BOTH genes against the reference's
    one-thread `search ... 1`:
      * MEGAGTA_CACHE_WINDOW=1 (seed j sees every seed before it) is the reference's sequential run: byte-identical FASTA;
      * the default mode is a window (here 16, far fewer than the seeds): seed j sees the seeds <= j - 16.  Deterministic, but NOT the
        sequential run: a few seeds take another of the equally scored paths.  The multiset of contigs is compared with `search ... 1`
        and must agree on all but a few per cent (the reference's own multi-thread runs differ from its one-thread run in the same way)."""
    _need()
    d = two_gene_inputs
    gl = str(d / "models" / "gene_list.txt")
    out = d / "out"
    env = {**os.environ, "MEGAGTA_CACHE_WINDOW": "16"}
    r = subprocess.run([sys.executable, DRIVER, "-r", str(d / "reads.fa"), "-g", gl, "-k", "30,36,45", "-o", str(out), "-t", "4",
                        "--min-contig-len", "150"], capture_output=True, text=True, env=env)
    assert r.returncode == 0, r.stderr + open(out / "log").read()[-2000:]
    res = H.stagewise_vs_reference(out, d, gl, oracle, REF, BIN)
    assert set(res) == {"rplB", "nirK"}
    for gene, (common_n, n, same) in res.items():
        assert same, (gene, common_n, n)                 # on this input the window changes no contig at all (deterministic: same every run)
        print(f"config3 {gene}: {common_n} of {n} raw contigs equal `search ... 1` as a multiset (window 16)")


def test_worker_process_and_one_process_per_step_write_identical_artefacts(toy_inputs):
    """the driver's default (ONE `megagta serve` worker: context, read library and the last graph stay resident between the steps) and
    `--one-process-per-step` (the reference driver's way: every step reads its inputs from the files) produce the same files, byte for
    byte: graphs, contigs of the intermediate k, seeds, raw and filtered contigs, checkpoints"""
    assert os.path.exists(BIN), "megagta_amd/bin/megagta missing: run __graft_entry__.build()"
    d = toy_inputs
    outs = []
    for tag, extra in (("w", []), ("p", ["--one-process-per-step"])):
        out = d / f"out_mode_{tag}"
        r = subprocess.run([sys.executable, DRIVER, "-r", str(d / "reads.fa"), "-g", str(d / "gene_list.txt"), "-k", "30,36,45", "-o", str(out),
                            "-t", "4", "--min-contig-len", "150", "--verbose"] + extra, capture_output=True, text=True)
        assert r.returncode == 0, r.stderr[-3000:]
        outs.append(out)
        log = (out / "log").read_text()
        assert ("still on the device" in log) == (tag == "w")          # the hand-off really happened (and only in the worker)
    files = ["k29/29.sdbg.0", "k29/29.sdbg_info", "k29/29.contigs.fa", "k35/35.sdbg.0", "k35/35.contigs.fa", "k35/35.contigs.fa.info",
             "k44/44.sdbg.0", "k44/44.sdbg_info", "k44/44_rplB_starting_kmers.txt", "k44/44_raw_contigs_rplB.fasta",
             "contigs/rplB/nucl_merged.fasta", "contigs/rplB/prot_merged.fasta", "tmp/cp.txt", "tmp/reads.lib.bin"]
    for f in files:
        a, b = (outs[0] / f).read_bytes(), (outs[1] / f).read_bytes()
        assert a == b and len(a) > 0, f
    # a failing step in the worker is reported like a failing child process
    bad = d / "gene_list_bad.txt"
    bad.write_text("rplB /nonexistent/for.hmm /nonexistent/rev.hmm " + (d / "gene_list.txt").read_text().split()[3] + "\n")
    r = subprocess.run([sys.executable, DRIVER, "-r", str(d / "reads.fa"), "-g", str(bad), "-k", "45", "-o", str(d / "out_bad"), "-t", "4"],
                       capture_output=True, text=True)
    assert r.returncode != 0 and "Error occurs when running" in r.stderr


def test_sharded_search_one_and_two_ranks_vs_megagta_search(two_gene_inputs):
    """`search_dist.py` (the search step of `megagta.py --gpus N`): with one rank its FASTA files are `megagta search`'s byte for byte;
    with two ranks (genes -> ranks first: rank 0 takes rplB, rank 1 nirK; gloo here, two processes on the one GPU of the box) every
    gene is searched by one rank over all its seeds, so the files are again identical; with the seeds of ONE gene split over two ranks
    (a one-gene list) every rank windows over its own half, and the multiset of contigs still equals the one-rank run on this input"""
    assert os.path.exists(BIN)
    d = two_gene_inputs
    out = d / "out"
    if not (out / "k44" / "44.sdbg_info").exists():
        pytest.skip("needs the driver run of test_config3_two_genes_multi_k_stagewise_and_contig_multiset")
    gl = str(d / "models" / "gene_list.txt")
    pre = str(out / "k44" / "44")
    env = {**os.environ, "MEGAGTA_CACHE_WINDOW": "16"}
    subprocess.run([BIN, "search", pre, gl, pre, str(d / "sd_ref"), "20", "0.5", "4"], check=True, capture_output=True, env=env)
    script = os.path.join(ROOT, "megagta_amd", "search_dist.py")
    r = subprocess.run([sys.executable, script, pre, gl, pre, str(d / "sd_w1"), "20", "0.5", "4"], capture_output=True, text=True, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    for gene in ("rplB", "nirK"):
        assert (d / f"sd_w1_raw_contigs_{gene}.fasta").read_bytes() == (d / f"sd_ref_raw_contigs_{gene}.fasta").read_bytes()

    def two_ranks(gene_list, tag):
        import socket
        s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
        e2 = {**env, "MEGAGTA_DIST_BACKEND": "gloo", "MEGAGTA_DEVICE": "0", "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port), "WORLD_SIZE": "2"}
        ps = [subprocess.Popen([sys.executable, script, pre, gene_list, pre, str(d / tag), "20", "0.5", "4"], env={**e2, "RANK": str(rk), "LOCAL_RANK": str(rk)},
                               stderr=subprocess.PIPE, text=True) for rk in range(2)]
        for p in ps:
            _, err = p.communicate(timeout=600)
            assert p.returncode == 0, err[-2000:]

    two_ranks(gl, "sd_w2")
    for gene in ("rplB", "nirK"):
        assert (d / f"sd_w2_raw_contigs_{gene}.fasta").read_bytes() == (d / f"sd_ref_raw_contigs_{gene}.fasta").read_bytes()
    one = d / "gene_list_rplB.txt"
    one.write_text(open(gl).readline())
    two_ranks(str(one), "sd_w2one")
    from collections import Counter
    a, b = _fasta_seqs(d / "sd_w2one_raw_contigs_rplB.fasta"), _fasta_seqs(d / "sd_ref_raw_contigs_rplB.fasta")
    assert len(a) == len(b) and Counter(a) == Counter(b)
    names = [l for l in open(d / "sd_w2one_raw_contigs_rplB.fasta") if l.startswith(">")]
    assert names == [l for l in open(d / "sd_ref_raw_contigs_rplB.fasta") if l.startswith(">")]
