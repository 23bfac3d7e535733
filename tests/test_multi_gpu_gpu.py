"""BASELINE configs 4-5 in small, on the one GPU of the box: the rank-aware product path (`megagta.py --gpus N`: buildgraph sharded by
prefix bucket into one .sdbg file per rank, graph files parsed on the device by every rank, seeds sharded by gene first, one all-gather
of contigs) against the one-GPU run.  Ranks share the device here (MEGAGTA_DEVICE=0) and talk over gloo (RCCL refuses two ranks per
device); the code path is the one `--gpus N` takes on an 8-GPU node."""
import os
import subprocess
import sys
from collections import Counter

import numpy as np
import pytest

from megagta_amd import synth
from tests import helpers as H

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.path.join(ROOT, "oracle", "_ref", "megagta")
BIN = os.path.join(ROOT, "megagta_amd", "bin", "megagta")
DRIVER = os.path.join(ROOT, "megagta_amd", "megagta.py")
ONE_GPU = {"MEGAGTA_DEVICE": "0", "MEGAGTA_DIST_BACKEND": "gloo"}


@pytest.fixture(scope="module")
def ctx():
    from megagta_amd import api
    c = api.Context(0)
    yield c
    c.close()


def _seqs(path):
    return [l for l in open(path).read().splitlines() if l and l[0] != ">"]


def test_graph_from_files_is_the_graph_from_the_stream(ctx, oracle, golden_dir, tmp_path):
    """mgta_sdbg_load_files (index parsed on the host, record files copied to the device and parsed there, one bucket per lane) ==
    mgta_sdbg_load of the decoded stream: every line of the graph, on graphs with tips, large multiplicities, empty buckets, one file
    and several files (the reference's writer deals the buckets to num_threads files)"""
    from megagta_amd import api, readlib
    packed, start = readlib.load_for_build(os.path.join(golden_dir, "ragged", "reads.lib"))
    for k in (29, 47):
        stream = ctx.build_sdbg(ctx.upload_reads(packed, start), k)
        assert stream.tips.size > 0 and stream.large.size > 0
        g0 = api.Graph(ctx, stream)
        edges = np.arange(g0.size, dtype=np.int64)
        want = g0.outgoing(edges)
        probe = ["".join("ACGT"[c] for c in np.random.default_rng(k + i).integers(0, 4, k + 1)) for i in range(64)]
        # (MGTA_LOAD_RANGE_RECORDS: the loader decodes the records range by range into one buffer and packs the lines each range completes -- a
        # graph of tens of billions of edges cannot hold its records beside its lines; tiny ranges put a range border inside nearly every line)
        for nf, rng in ((1, None), (3, None), (7, None), (1, "64"), (3, "1000"), (7, "70001"), (3, "200")):
            pre = str(tmp_path / f"g{k}_{nf}")
            api.write_sdbg(pre, stream, num_files=nf)
            if rng:
                os.environ["MGTA_LOAD_RANGE_RECORDS"] = rng
            try:
                g = api.Graph.from_files(ctx, pre)
            finally:
                os.environ.pop("MGTA_LOAD_RANGE_RECORDS", None)
            assert g.size == g0.size and g.k == k
            got = g.outgoing(edges)
            assert np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1]), (nf, rng)
            assert np.array_equal(g.invalid_bits(), g0.invalid_bits())
            assert np.array_equal(g.index_edges(probe), g0.index_edges(probe))
            g.free()
        g0.free()
    # a damaged index is an error, not a crash: sizes that do not add up, a bucket beyond the end of its file
    pre = str(tmp_path / "g47_1")
    info = open(pre + ".sdbg_info").read().splitlines()
    first = next(i for i, l in enumerate(info) if len(l.split()) == 6 and l.split()[1] != "-1")
    t = info[first].split()
    for bad in (" ".join(t[:3] + [str(int(t[3]) + 1)] + t[4:]), " ".join(t[:2] + [str(10 ** 9)] + t[3:])):
        open(pre + "_bad.sdbg_info", "w").write("\n".join(info[:first] + [bad] + info[first + 1:]) + "\n")
        if os.path.lexists(pre + "_bad.sdbg.0"):
            os.remove(pre + "_bad.sdbg.0")
        os.symlink(pre + ".sdbg.0", pre + "_bad.sdbg.0")
        with pytest.raises(api.MegaGtaError):
            api.Graph.from_files(ctx, pre + "_bad")


@pytest.mark.parametrize("extra,tag", [([], "m1"), (["-m", "2", "--need_mercy"], "m2")])
def test_buildgraph_over_three_ranks_writes_the_one_rank_graph(oracle, golden_dir, tmp_path, extra, tag):
    """`megagta buildgraph` as three ranks (MEGAGTA_RANK / MEGAGTA_WORLD): each builds its share of the prefix buckets into PREFIX.sdbg.<r>,
    `sdbgmerge` writes the index: the logical edge stream is the one-process stream (also with -m 2 --need_mercy: stage 1 runs whole on
    every rank), and the reference's own reader takes the three files (`denovo -t 1` on them == on the one-file graph)"""
    assert os.path.exists(BIN)
    lib = os.path.join(golden_dir, "ragged", "reads.lib")
    common = ["-k", "29", "--host_mem", "4000000000", "--mem_flag", "1", "--gpu_mem", "0", "--num_cpu_threads", "4", "--num_output_threads", "1",
              "--read_lib_file", lib] + (extra if extra else ["-m", "1"])
    run = lambda cmd, **kw: subprocess.run(cmd, check=True, capture_output=True, **kw)
    run([BIN, "buildgraph", "--output_prefix", str(tmp_path / "one")] + common)
    ps = [subprocess.Popen([BIN, "buildgraph", "--output_prefix", str(tmp_path / "three")] + common, stderr=subprocess.PIPE,
                           env={**os.environ, "MEGAGTA_RANK": str(r), "MEGAGTA_WORLD": "3", "MEGAGTA_DEVICE": "0"}) for r in range(3)]
    for p in ps:
        _, err = p.communicate(timeout=600)
        assert p.returncode == 0, err[-1500:]
    run([BIN, "sdbgmerge", str(tmp_path / "three"), "3"])
    a, b = oracle.Stream.read(str(tmp_path / "one")).edges(), oracle.Stream.read(str(tmp_path / "three")).edges()
    assert a.md5() == b.md5() and a.records.size > 1000
    assert all(os.path.getsize(tmp_path / f"three.sdbg.{r}") > 0 for r in range(3)) and not os.path.exists(tmp_path / "three.sdbg_info.part0")
    if extra:
        assert (tmp_path / "one.counting").read_text() == (tmp_path / "three.counting").read_text()
    if os.path.exists(REF):
        for g in ("one", "three"):
            run([REF, "denovo", "-s", str(tmp_path / g), "-o", str(tmp_path / g), "-t", "1", "--max_tip_len", "60", "--min_contig", "31"])
        assert (tmp_path / "one.contigs.fa").read_bytes() == (tmp_path / "three.contigs.fa").read_bytes()
    run([BIN, "denovo", "-s", str(tmp_path / "three"), "-o", str(tmp_path / "ours3"), "--max_tip_len", "60", "--min_contig", "31"])
    if os.path.exists(REF):
        assert (tmp_path / "ours3.contigs.fa").read_bytes() == (tmp_path / "three.contigs.fa").read_bytes()


@pytest.fixture(scope="module")
def five_gene_inputs(tmp_path_factory):
    """BASELINE.json configs[3] in small: five genes of different lengths, every genome carries one diverged copy of each"""
    d = tmp_path_factory.mktemp("cfg4")
    mg = synth.make_metagenome(16000, 150, (("rplB", 277), ("nirK", 360), ("nifH", 296), ("rpoB", 240), ("amoA", 180)), seed=31,
                               reads_per_genome=1000, genome_len=14000)
    synth.write_fasta(mg.reads, str(d / "reads.fa"))
    synth.write_gene_models(mg.genes, str(d / "models"))
    return d


def test_config4_five_genes_two_ranks_every_artefact_equals_the_one_gpu_run(five_gene_inputs):
    """`megagta.py -k 30,36,45 --gpus 2` with a five-gene list against `--gpus 1`: the three graphs (each written as two files by two
    ranks) decode to the same streams, the contigs of the intermediate k and the seeds are byte-identical, and -- fewer ranks than genes,
    so whole genes are dealt to the ranks and every gene is searched by one rank over all its seeds -- so are the raw and the filtered
    contigs of all five genes"""
    assert os.path.exists(BIN)
    from oracle import oracle as O
    O.build()
    d = five_gene_inputs
    gl = str(d / "models" / "gene_list.txt")
    env = {**os.environ, "MEGAGTA_CACHE_WINDOW": "16"}
    outs = {}
    started = None
    for gpus in (1, 2):
        out = d / f"out{gpus}"
        r = subprocess.run([sys.executable, DRIVER, "-r", str(d / "reads.fa"), "-g", gl, "-k", "30,36,45", "-o", str(out), "-t", "4", "--min-contig-len", "150",
                            "--gpus", str(gpus), "--verbose"], capture_output=True, text=True, env={**env, **(ONE_GPU if gpus > 1 else {})})
        assert r.returncode == 0, r.stderr[-3000:] + open(out / "log").read()[-3000:]
        outs[gpus] = out
        if gpus == 1 and os.path.exists(REF):
            # the reference's steps on the one-GPU run's artefacts (host threads, one GPU process) go on BESIDE the two-rank run below (the test
            # runner, the driver's worker, two ranks and this one search: five processes on the card, the box admits six)
            started = H.stagewise_start(outs[1], d, gl, REF, BIN)
    genes = [l.split()[0] for l in open(gl)]
    assert len(genes) == 5
    for k in (29, 35, 44):
        assert os.path.exists(outs[2] / f"k{k}" / f"{k}.sdbg.1") and not os.path.exists(outs[1] / f"k{k}" / f"{k}.sdbg.1")
        a, b = O.Stream.read(str(outs[1] / f"k{k}" / f"{k}")).edges(), O.Stream.read(str(outs[2] / f"k{k}" / f"{k}")).edges()
        assert a.md5() == b.md5() and a.records.size > 100000, k
    files = ["k29/29.contigs.fa", "k35/35.contigs.fa", "k35/35.contigs.fa.info"]
    for g in genes:
        files += [f"k44/44_{g}_starting_kmers.txt", f"k44/44_raw_contigs_{g}.fasta", f"contigs/{g}/nucl_merged.fasta", f"contigs/{g}/prot_merged.fasta"]
    for f in files:
        a, b = (outs[1] / f).read_bytes(), (outs[2] / f).read_bytes()
        assert a == b and len(a) > 0, f
    log = (outs[2] / "log").read_text()
    assert "on 2 GPUs" in log and "rank 1 of 2" in log
    # ... and the one-GPU run against the REFERENCE binary, stage by stage (config 4 in small: the gene loop of search.cpp:105-122 over five
    # genes): three graphs, two sets of contigs, five seed files, five raw-contig files (window 1 == `search ... 1` byte for byte; the
    # driver's window-16 run as a multiset)
    if started is None:
        pytest.skip("oracle/_ref/megagta (the prebuilt reference) is not present: the reference half of the test")
    res = H.stagewise_finish(started, O)
    assert set(res) == set(genes)
    print("config4 vs reference: " + ", ".join(f"{g} {c}/{n}" for g, (c, n, _) in res.items()) + " raw contigs equal `search ... 1` as a multiset (window 16)")


RCCL_WORLD1 = r"""
import os, sys, numpy as np, torch, torch.distributed as dist
sys.path.insert(0, sys.argv[1])
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", sys.argv[2])
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
from megagta_amd import dist as mdist
assert dist.get_backend() == "nccl" and mdist._dev().type == "cuda"
rng = np.random.default_rng(3)
recs = torch.from_numpy(rng.integers(0, 65536, 100001, dtype=np.uint16).view(np.uint8).copy()).cuda()
whole = mdist.all_gather_record_shards(recs)
assert whole.is_cuda and torch.equal(whole, recs)                       # device to device: the shard never visits the host
n_seeds = [0, 5, 1000]
mine = [np.arange(n, dtype=np.int64) for n in n_seeds]
blobs = [[(b"acgt" * ((i * 7 + g) % 40))[: (i * 13 + g) % 150] for i in range(n)] for g, n in enumerate(n_seeds)]
offs = [np.concatenate([[0], np.cumsum([len(b) for b in bl])]).astype(np.int64) for bl in blobs]
cont = [np.frombuffer(b"".join(bl), dtype=np.uint8) for bl in blobs]
merged = mdist.all_gather_all_genes(n_seeds, mine, cont, offs)
for g, (c, o) in enumerate(merged):
    assert [x.encode() for x in mdist.contig_list(c, o)] == blobs[g], g
c, o = mdist.all_gather_packed_contigs(n_seeds[2], mine[2], cont[2], offs[2])
assert np.array_equal(c, cont[2]) and np.array_equal(o, offs[2])
dist.barrier()
dist.destroy_process_group()
print("RCCL world 1 ok")
"""


def test_rccl_backend_world_one_runs_the_two_exchanges_on_device_tensors(tmp_path):
    """the `nccl` (= RCCL) branch of megagta_amd/dist.py has only ever been taken by the driver's 8-GPU run: here the process group is RCCL
    with ONE rank on the box's GPU, and both exchanges of the path -- the record shards of a sharded build, the contigs of all genes at
    the end of a sharded search -- run on device tensors through `all_gather_into_tensor`"""
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    script = tmp_path / "rccl1.py"
    script.write_text(RCCL_WORLD1)
    r = subprocess.run([sys.executable, str(script), ROOT, str(port)], capture_output=True, text=True, timeout=600,
                       env={**os.environ, "HSA_ENABLE_IPC_MODE_LEGACY": "0"})
    assert r.returncode == 0 and "RCCL world 1 ok" in r.stdout, r.stderr[-3000:]


def test_config5_ten_genes_over_three_ranks_on_the_hip_path(tmp_path):
    """BASELINE.json configs[4] in small, on the HIP path: ten genes, `search_dist.py` as THREE ranks sharing the one GPU over gloo (a box
    lets at most six processes use its card at once -- the test runner's own context counts --, so not eight; tests/test_dist_gloo.py runs
    the ten-gene partition over 8 and 4 ranks on the CPU).
    Fewer ranks than genes: whole genes are dealt to the ranks (heaviest first), every gene is searched by one rank over all its seeds in
    order, ONE all-gather brings the contigs to rank 0.  With window 1, against `megagta search` on the same files and against the REFERENCE's
    `search ... 1` (the gene loop of search.cpp:105-122): every gene's FASTA byte-identical."""
    assert os.path.exists(BIN)
    d = tmp_path
    specs = (("rplB", 277), ("nirK", 360), ("nifH", 296), ("rpoB", 240), ("amoA", 180), ("nosZ", 200), ("pmoA", 150), ("dsrA", 220), ("mcrA", 260), ("nxrB", 170))
    mg = synth.make_metagenome(15000, 150, specs, seed=41, reads_per_genome=1500, genome_len=22000)
    synth.write_lib_bin(mg.reads, str(d / "reads.lib"))
    gl = synth.write_gene_models(mg.genes, str(d / "models"))
    run = lambda cmd, **kw: subprocess.run(cmd, check=True, capture_output=True, **kw)
    run([BIN, "buildgraph", "-k", "44", "-m", "1", "--host_mem", "4000000000", "--mem_flag", "1", "--gpu_mem", "0", "--num_cpu_threads", "4",
         "--num_output_threads", "1", "--read_lib_file", str(d / "reads.lib"), "--output_prefix", str(d / "44")])
    genes = {l.split()[0]: l.split()[3] for l in open(gl)}
    assert len(genes) == 10
    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(max_workers=4) as ex:                     # (four one-shot processes at a time on the card: the box admits six)
        seed_lines = dict(zip(genes, ex.map(lambda faa: run([BIN, "findstart", faa, str(d / "reads.lib.bin"), "45", "4"]).stdout.splitlines(keepends=True), genes.values())))
    for g, lines in seed_lines.items():
        assert len(lines) > 100, g
        with open(d / f"44_{g}_starting_kmers.txt", "wb") as f:     # (window 1 = one search at a time per direction: the first 80 seeds of every gene)
            f.write(b"".join(lines[:80]))
    pre = str(d / "44")
    ref_run = subprocess.Popen([REF, "search", pre, gl, pre, str(d / "ref1"), "20", "0.5", "1"], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL) \
        if os.path.exists(REF) else None                              # (the reference's one-thread run, behind ours)
    script = os.path.join(ROOT, "megagta_amd", "search_dist.py")
    # (one launch of the ranks: every process pays ~a minute of `import torch` on a fresh box.  Window 1 has the reference to compare with;
    # the default mode over ranks against `megagta search` is test_process_boundary_gpu.py::test_sharded_search_one_and_two_ranks_...)
    for tag, env in (("w1", {**os.environ, "MEGAGTA_CACHE_WINDOW": "1"}),):
        run([BIN, "search", pre, gl, pre, str(d / f"one_{tag}"), "20", "0.5", "4"], env=env)
        r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--standalone", "--local-addr", "127.0.0.1", "--nnodes=1", "--nproc-per-node", "3",
                            script, pre, gl, pre, str(d / f"three_{tag}"), "20", "0.5", "4"], capture_output=True, text=True, env={**env, **ONE_GPU})
        assert r.returncode == 0, r.stderr[-3000:]
        for g in genes:
            a, b = (d / f"one_{tag}_raw_contigs_{g}.fasta").read_bytes(), (d / f"three_{tag}_raw_contigs_{g}.fasta").read_bytes()
            assert a == b and a.count(b">") > 20, (tag, g)
    if ref_run is not None:
        assert ref_run.wait(timeout=600) == 0
        for g in genes:
            assert (d / f"three_w1_raw_contigs_{g}.fasta").read_bytes() == (d / f"ref1_raw_contigs_{g}.fasta").read_bytes(), g
        print("config5 in small: ten genes over three ranks == `megagta search` == the reference's `search ... 1` (window 1)")


def test_split_gene_agreement_fraction(tmp_path):
    """more ranks than genes: the seeds of ONE gene are dealt round-robin to two ranks, each windows over its own half.  The result is
    deterministic for a given (seed order, N, B) but NOT the one-rank result.  Input chosen (scripts/explore_window_effects.py) so that
    sharing really matters: 4x coverage, 2 % errors, prune 10 -- cold and sequential runs differ on 166 of 9427 seeds there, and the
    two-rank run differs from the one-rank run on about 1 % of the seeds.  How far apart they are is measured, printed and bounded."""
    assert os.path.exists(BIN)
    d = tmp_path
    mg = synth.make_metagenome(20000, 150, (("g", 277),), seed=6, reads_per_genome=300, genome_len=12000, aa_sub=0.03, err=0.02)
    synth.write_lib_bin(mg.reads, str(d / "reads.lib"))
    gl = synth.write_gene_models(mg.genes, str(d / "models"))
    run = lambda cmd, **kw: subprocess.run(cmd, check=True, capture_output=True, **kw)
    run([BIN, "buildgraph", "-k", "44", "-m", "1", "--host_mem", "4000000000", "--mem_flag", "1", "--gpu_mem", "0", "--num_cpu_threads", "4",
         "--num_output_threads", "1", "--read_lib_file", str(d / "reads.lib"), "--output_prefix", str(d / "44")])
    faa = open(gl).readline().split()[3]
    with open(d / "44_g_starting_kmers.txt", "wb") as f:
        f.write(run([BIN, "findstart", faa, str(d / "reads.lib.bin"), "45", "4"]).stdout)
    pre = str(d / "44")
    env = {**os.environ, "MEGAGTA_CACHE_WINDOW": "8", "MEGAGTA_CACHE_COST_RATE": "0"}
    run([BIN, "search", pre, gl, pre, str(d / "s1"), "10", "0.5", "4"], env=env)
    run([BIN, "search", pre, gl, pre, str(d / "cold"), "10", "0.5", "4"], env={**env, "MEGAGTA_CACHE_WINDOW": "0"})
    script = os.path.join(ROOT, "megagta_amd", "search_dist.py")
    runs = []
    for rep in range(1):                                              # (that the result is the same on every run is what the driver-run comparisons above rest on)
        r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--standalone", "--local-addr", "127.0.0.1", "--nnodes=1", "--nproc-per-node", "2",
                            script, pre, gl, pre, str(d / f"s2_{rep}"), "10", "0.5", "4"], capture_output=True, text=True, env={**env, **ONE_GPU})
        assert r.returncode == 0, r.stderr[-3000:]
        runs.append((d / f"s2_{rep}_raw_contigs_g.fasta").read_bytes())
    a, b, c = _seqs(d / "s2_0_raw_contigs_g.fasta"), _seqs(d / "s1_raw_contigs_g.fasta"), _seqs(d / "cold_raw_contigs_g.fasta")
    assert len(a) == len(b) == len(c) > 5000
    differ = sum(1 for x, y in zip(a, b) if x != y)
    common = sum((Counter(a) & Counter(b)).values())
    cold_differ = sum(1 for x, y in zip(c, b) if x != y)
    print(f"split gene: {len(a)} seeds over 2 ranks, window 8: {differ} contigs differ from the one-rank run seed by seed ({100.0 * differ / len(a):.2f} %), "
          f"{common} equal as a multiset ({100.0 * common / len(a):.2f} %); no sharing at all differs on {cold_differ}")
    assert cold_differ > 0 and differ > 0, "the input was chosen because sharing changes contigs on it"
    assert common >= 0.97 * len(b), (common, len(b))
    names = [l for l in open(d / "s2_0_raw_contigs_g.fasta") if l.startswith(">")]
    assert names == [l for l in open(d / "s1_raw_contigs_g.fasta") if l.startswith(">")]


def test_bench_two_ranks_gathers_the_one_rank_stream():
    """the driver's multi-GPU call of bench.py (`torch.distributed.run --nproc-per-node N bench.py --gpus N`) rehearsed with two ranks on the
    one GPU over gloo: the bucket-sharded build + the variable-length all-gather of the record shards leave on every rank the stream the
    one-rank run builds (md5 of the records in bucket order), and the line keeps its contract (n_gpus, strong scaling, the rank-invariant
    floor next to the number)"""
    import json
    small = ["--reads", "200000", "--steps", "2", "--warmup", "1", "--seeds", "300", "--product-seeds", "0", "--e2e-reads", "0", "--no-cpu-baseline", "--check-stream"]
    env = {**os.environ, **ONE_GPU}
    one = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1"] + small, capture_output=True, text=True, env=env, timeout=600)
    assert one.returncode == 0, one.stderr[-2000:]
    two = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--standalone", "--local-addr", "127.0.0.1", "--nnodes=1", "--nproc-per-node", "2",
                          os.path.join(ROOT, "bench.py"), "--gpus", "2"] + small, capture_output=True, text=True, env=env, timeout=600)
    assert two.returncode == 0, two.stderr[-2000:]
    l1 = json.loads([l for l in one.stdout.splitlines() if l.startswith("{")][-1])
    l2 = json.loads([l for l in two.stdout.splitlines() if l.startswith("{")][-1])
    assert l1["n_gpus"] == 1 and l2["n_gpus"] == 2 and l2["scaling"] == "strong" and l2["config"]["reads"] == 200000
    assert l1["stream_md5"] == l2["stream_md5"] and len(l1["stream_md5"]) == 32
    assert l2["whole_build"]["rank_invariant_floor_ms"] > 0 and l2["value"] > 0
    assert l2["search"]["expansions_per_step"] == l1["search"]["expansions_per_step"] > 0      # cold searches: the same work however the seeds are dealt
