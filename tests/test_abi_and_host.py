"""CPU-only: the C-ABI library loads and exports every declared symbol; host file formats round-trip."""
import ctypes
import os
import re

import numpy as np
import pytest

from megagta_amd import _lib, api, readlib, synth
from tests import helpers as H

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "megagta_hip.h")).read()
    declared = set(re.findall(r"\b(mgta_[a-z_0-9]+)\s*\(", hdr)) - {"mgta_edge_sink", "mgta_contig_sink"}
    assert declared == set(_lib.SYMBOLS), declared ^ set(_lib.SYMBOLS)
    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    L = ctypes.CDLL(_lib.LIB_PATH)
    for name in declared:
        assert hasattr(L, name), name


def test_no_cpu_fallback_without_device():
    """no GPU here: creating a context must fail loudly (never a silent CPU path)"""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(api.MegaGtaError):
        api.Context(0)


def test_product_never_imports_oracle():
    """the oracle is test infrastructure: nothing under megagta_amd/ may import, link or call it"""
    pat = re.compile(r"import\s+oracle|from\s+oracle|liboracle|mgta_oracle\.h|\borc_[a-z_]+\s*\(")
    for dirpath, _, files in os.walk(os.path.join(ROOT, "megagta_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".cpp", ".h")) or f == "Makefile":
                assert not pat.search(open(os.path.join(dirpath, f)).read()), f


def test_lib_bin_roundtrip(tmp_path, golden_dir):
    mg = synth.make_metagenome(500, 150, (("rplB", 60),), seed=2)
    synth.write_lib_bin(mg.reads, str(tmp_path / "r.lib"))
    reads = readlib.load_lib_bin(str(tmp_path / "r.lib"))
    assert len(reads) == 500 and all(np.array_equal(a, b) for a, b in zip(reads, mg.reads))
    p1, s1 = readlib.pack_for_build(reads)
    p2, s2 = synth.pack_reads_for_build(mg.reads)
    assert np.array_equal(p1, p2) and np.array_equal(s1, s2)
    # the committed golden library (written by the reference's buildlib) decodes to 6000 x 150
    g = readlib.load_lib_bin(os.path.join(golden_dir, "toy", "reads.lib"))
    assert len(g) == 6000 and all(r.size == 150 for r in g)


def test_sdbg_files_roundtrip(tmp_path, oracle, golden_dir):
    """write_sdbg output is read back identically by read_sdbg AND by the oracle's reader (reference format)"""
    packed, start = readlib.load_for_build(os.path.join(golden_dir, "ragged", "reads.lib"))
    o = oracle.Stream.build(packed, start, 29, threads=2).edges()
    s = api.EdgeStream(k=o.k, words_per_tip=o.words_per_tip, bucket_items=o.bucket_items, records=o.records, large=o.large, tips=o.tips)
    for nf in (1, 3):
        prefix = str(tmp_path / f"g{nf}")
        api.write_sdbg(prefix, s, num_files=nf)
        back = api.read_sdbg(prefix)
        assert back.md5() == s.md5()
        assert oracle.Stream.read(prefix).edges().md5() == s.md5()


def _plan(n_items, W, b0, b1, mode=None):
    L = _lib.load()
    old = os.environ.get("MGTA_SORT_BIAS")
    try:
        if mode is None:
            os.environ.pop("MGTA_SORT_BIAS", None)
        else:
            os.environ["MGTA_SORT_BIAS"] = str(mode)
        P, s = ctypes.c_int(), ctypes.c_int()
        _lib.check(L.mgta_sort_plan(n_items, W, b0, b1, ctypes.byref(P), ctypes.byref(s)), "mgta_sort_plan")
        return P.value, s.value
    finally:
        os.environ.pop("MGTA_SORT_BIAS", None)
        if old is not None:
            os.environ["MGTA_SORT_BIAS"] = old


def test_sort_plan_of_sub_range_builds():
    """host logic of the global sort passes (no device): the digits of a bucket sub-range build skip the leading bits its keys share, but
    only where that saves a pass and leaves short segments; the measured cases of DESIGN.md §5"""
    assert _plan(2_160_000_000, 3, 0, 65536) == (3, 0)                       # 10 M reads, one pass over every bucket
    assert _plan(2_160_000_000, 3, 0, 65536, mode=2) == (3, 0)               # nothing to skip in a whole-range build
    third = (65536 + 2) // 3
    assert _plan(7_200_000_000, 3, 0, third) == (4, 0)                       # 100 M reads, full memory: 645-key segments are not worth a pass
    assert _plan(7_200_000_000, 3, 0, third, mode=2) == (3, 1)
    assert _plan(7_200_000_000, 3, 0, third, mode=0) == (4, 0)
    w11 = (65536 + 10) // 11
    assert _plan(1_963_636_363, 3, 3 * w11, 4 * w11) == (3, 3)               # 100 M reads under 64 GB: 11 ranges, 161-key segments
    assert _plan(1_963_636_363, 3, 3 * w11, 4 * w11, mode=0) == (4, 0)
    assert _plan(185_000, 3, 777, 778) == (2, 16)                            # one crowded bucket: two digits below the 16 bucket bits
    assert _plan(185_000, 3, 777, 778, mode=0) == (4, 0)
    assert _plan(100, 3, 0, 65536) == (0, 0)
    rng = np.random.default_rng(11)
    for _ in range(3000):                                                    # invariants: prefix of 16..32 bits whenever bits are skipped
        b0 = int(rng.integers(0, 65536))
        b1 = int(rng.integers(b0 + 1, 65537))
        n = int(10 ** rng.uniform(1, 10.5))
        W = int(rng.integers(2, 10))
        for mode in (0, 1, 2):
            P, s = _plan(n, W, b0, b1, mode)
            assert 0 <= P <= 4 and 0 <= s <= 24
            if mode == 0 or (b0 == 0 and b1 == 65536):
                assert s == 0
            if s:
                assert 16 <= 8 * P + s <= 32 and ((b1 - b0) << 16) - 1 < (1 << (32 - s))
        assert _plan(n, W, b0, b1, 1)[0] <= _plan(n, W, b0, b1, 0)[0]        # the bias never adds a pass


def test_sort_plan_rejects_bad_arguments():
    L = _lib.load()
    P, s = ctypes.c_int(), ctypes.c_int()
    assert L.mgta_sort_plan(10, 3, 5, 5, ctypes.byref(P), ctypes.byref(s)) != 0
    assert L.mgta_sort_plan(10, 3, 0, 70000, ctypes.byref(P), ctypes.byref(s)) != 0
    assert L.mgta_sort_plan(10, 1, 0, 65536, ctypes.byref(P), ctypes.byref(s)) != 0


def test_host_sdbg_reader_and_writer(tmp_path, oracle, golden_dir):
    """the C++ graph-file reader and writer of bin/megagta (buildgraph writes, denovo / search read): a 3-file graph goes through
    `megagta sdbgcopy` and comes back as the same stream, for our Python reader and for the oracle's reader of the reference format"""
    import subprocess
    packed, start = readlib.load_for_build(os.path.join(golden_dir, "ragged", "reads.lib"))
    o = oracle.Stream.build(packed, start, 29, threads=2).edges()
    s = api.EdgeStream(k=o.k, words_per_tip=o.words_per_tip, bucket_items=o.bucket_items, records=o.records, large=o.large, tips=o.tips)
    assert s.large.size > 0 and s.tips.size > 0                      # multiplicities > 254 and tip labels are in the file
    api.write_sdbg(str(tmp_path / "in"), s, num_files=3)
    exe = os.path.join(ROOT, "megagta_amd", "bin", "megagta")
    r = subprocess.run([exe, "sdbgcopy", str(tmp_path / "in"), str(tmp_path / "out")], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert api.read_sdbg(str(tmp_path / "out")).md5() == s.md5()
    assert oracle.Stream.read(str(tmp_path / "out")).edges().md5() == s.md5()


def test_sdbgmerge_puts_the_ranks_index_parts_together(tmp_path, oracle, golden_dir):
    """`megagta sdbgmerge` (host only): the per-rank index parts a multi-GPU `buildgraph` leaves (PREFIX.sdbg_info.part<r>: one header line
    "k words_per_tip b_lo b_hi records tips large", then the rank's bucket lines) become ONE PREFIX.sdbg_info naming the ranks' files; the
    merged graph reads back as the whole stream through the C++ reader, our Python reader and the oracle's reader of the reference format.
    Parts that do not tile the 65536 buckets are an error."""
    import subprocess
    packed, start = readlib.load_for_build(os.path.join(golden_dir, "ragged", "reads.lib"))
    o = oracle.Stream.build(packed, start, 29, threads=2).edges()
    s = api.EdgeStream(k=o.k, words_per_tip=o.words_per_tip, bucket_items=o.bucket_items, records=o.records, large=o.large, tips=o.tips)
    ranks = 3
    api.write_sdbg(str(tmp_path / "w"), s, num_files=ranks)           # the same layout: file r holds a contiguous bucket range
    info = open(tmp_path / "w.sdbg_info").read().splitlines()
    rows = [l.split() for l in info[7:]]
    assert len(rows) == 65536
    file_of = [int(r[1]) for r in rows]
    cuts = [0] + [next(b for b in range(65536) if file_of[b] == f) if f in file_of else None for f in range(1, ranks)] + [65536]
    # (empty buckets carry file -1: a part's range runs from its first bucket to the next part's first)
    assert all(c is not None for c in cuts)
    for r in range(ranks):
        os.replace(tmp_path / f"w.sdbg.{r}", tmp_path / f"m.sdbg.{r}")
        lo, hi = cuts[r], cuts[r + 1]
        mine = rows[lo:hi]
        n = sum(int(x[3]) for x in mine); nt = sum(int(x[4]) for x in mine); nl = sum(int(x[5]) for x in mine)
        with open(tmp_path / f"m.sdbg_info.part{r}", "w") as f:
            f.write(f"{s.k} {s.words_per_tip} {lo} {hi} {n} {nt} {nl}\n")
            f.writelines(" ".join(x) + "\n" for x in mine)
    exe = os.path.join(ROOT, "megagta_amd", "bin", "megagta")
    r = subprocess.run([exe, "sdbgmerge", str(tmp_path / "m"), str(ranks)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert not os.path.exists(tmp_path / "m.sdbg_info.part0")
    assert open(tmp_path / "m.sdbg_info").read().splitlines() == info
    assert api.read_sdbg(str(tmp_path / "m")).md5() == s.md5()
    assert oracle.Stream.read(str(tmp_path / "m")).edges().md5() == s.md5()
    r = subprocess.run([exe, "sdbgcopy", str(tmp_path / "m"), str(tmp_path / "c")], capture_output=True, text=True)
    assert r.returncode == 0 and api.read_sdbg(str(tmp_path / "c")).md5() == s.md5()
    # a missing rank is reported, not papered over
    r = subprocess.run([exe, "sdbgmerge", str(tmp_path / "m"), str(ranks)], capture_output=True, text=True)
    assert r.returncode != 0 and "part0" in r.stderr


def test_driver_continue_mode_follows_the_reference(tmp_path, monkeypatch, capsys):
    """--continue (reference megagta.py:321-351): every option but -o comes from opts.txt, parsed into a FRESH option set (reads given
    next to --continue are not appended a second time); without an opts.txt the driver says so and carries on in normal mode"""
    import importlib
    from megagta_amd import megagta as drv
    drv = importlib.reload(drv)
    out = tmp_path / "run"
    (out / "tmp").mkdir(parents=True)
    (out / "opts.txt").write_text("-r\nreads.fa\n-g\ngenes.txt\n-k\n30,45\n-o\n" + str(out) + "\n")
    (out / "tmp" / "cp.txt").write_text("0\tdone\n1\tdone\n2\tdone\n")
    drv.parse_opt(["-r", "other.fa", "--continue", "-o", str(out)])
    assert drv.opt.continue_mode and drv.opt.last_cp == 2
    assert drv.opt.se == ["reads.fa"] and drv.opt.k_list == [30, 45] and drv.opt.gene_list == "genes.txt"
    drv = importlib.reload(drv)
    empty = tmp_path / "nothing"
    empty.mkdir()
    drv.parse_opt(["-r", "x.fa", "--continue", "-o", str(empty)])
    assert not drv.opt.continue_mode and drv.opt.se == ["x.fa"] and "switching to normal mode" in capsys.readouterr().err


def test_driver_checkpoint_order_of_the_last_step(tmp_path, monkeypatch):
    """search_contigs writes the checkpoints of filterbylen / translate of every gene INSIDE the search step and the search's own last
    (reference megagta.py:680-760), so that either driver can continue the other's run"""
    import importlib
    from megagta_amd import megagta as drv
    drv = importlib.reload(drv)
    calls = []
    monkeypatch.setattr(drv, "run_step", lambda cmd, what, stdin_path=None, stdout_path=None: calls.append(cmd[1]))
    drv.opt.out_dir = str(tmp_path) + "/"
    drv.opt.temp_dir = drv.opt.out_dir + "tmp/"
    os.makedirs(drv.opt.temp_dir)
    drv.opt.gene_info = {"rplB": ("f", "r", "a"), "nirK": ("f", "r", "a")}
    drv.search_contigs(44)
    assert calls == ["search", "filterbylen", "translate", "filterbylen", "translate"]
    assert open(drv.opt.temp_dir + "cp.txt").read() == "".join(f"{i}\tdone\n" for i in range(5))
    # continuing a run whose search checkpoint is there: nothing runs, ONE checkpoint is passed (the nested ones are skipped with it)
    drv = importlib.reload(drv)
    monkeypatch.setattr(drv, "run_step", lambda *a, **k: calls.append("again"))
    drv.opt.out_dir = str(tmp_path) + "/"
    drv.opt.temp_dir = drv.opt.out_dir + "tmp/"
    drv.opt.gene_info = {"rplB": ("f", "r", "a"), "nirK": ("f", "r", "a")}
    drv.opt.continue_mode, drv.opt.last_cp = True, 4
    drv.search_contigs(44)
    assert "again" not in calls and drv.cp == 1


def test_side_by_side_filters_fail_the_run_when_a_chain_throws(tmp_path, monkeypatch):
    """the per-gene filter chains run in threads: an exception inside one (here: the raw contigs of one gene do not exist) must end the
    run as a failed step -- not leave the gene without result lines, skip its checkpoints and exit 0 (advisor r5)"""
    import importlib
    import pytest
    from megagta_amd import megagta as drv
    drv = importlib.reload(drv)
    drv.opt.out_dir = str(tmp_path) + "/"
    drv.opt.temp_dir = drv.opt.out_dir + "tmp/"
    os.makedirs(drv.opt.temp_dir)
    drv.opt.gene_info = {"rplB": ("f", "r", "a"), "nirK": ("f", "r", "a")}
    drv.opt.bin = "/bin/cat"                               # `cat filterbylen 250 < raw > nucl`: fails on its arguments, like any bad step
    os.makedirs(os.path.dirname(drv.graph_prefix(44)))
    open(drv.graph_prefix(44) + "_raw_contigs_rplB.fasta", "w").write(">a\nACGT\n")   # nirK's file is missing: open() raises in its thread
    with pytest.raises(SystemExit) as e:
        drv.filter_and_translate_side_by_side(44)
    assert e.value.code != 0


def test_search_plan_is_one_table_for_the_binary_and_the_ranks(monkeypatch):
    """the ordered-commit window and the cost term are chosen by the number of seeds of a gene's batch: `megagta search` (C++,
    `megagta searchplan N...` prints its choice, host only) and the multi-GPU ranks (search_dist.window_and_rate) must agree, with and
    without the MEGAGTA_CACHE_WINDOW / MEGAGTA_CACHE_COST_RATE overrides"""
    import subprocess
    from megagta_amd import search_dist
    exe = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "megagta_amd", "bin", "megagta")
    ns = [0, 1, 7000, 32767, 32768, 65535, 65536, 196607, 196608, 393215, 393216, 1_000_000, 9_300_000]
    for env in ({}, {"MEGAGTA_CACHE_WINDOW": "1"}, {"MEGAGTA_CACHE_WINDOW": "-1"}, {"MEGAGTA_CACHE_WINDOW": "64", "MEGAGTA_CACHE_COST_RATE": "-2"},
                {"MEGAGTA_CACHE_COST_RATE": "0"}, {"MEGAGTA_CACHE_WINDOW": "1", "MEGAGTA_CACHE_COST_RATE": "3"},
                {"MEGAGTA_CACHE_WINDOW": "", "MEGAGTA_CACHE_COST_RATE": ""}, {"MEGAGTA_CACHE_WINDOW": "-3"}, {"MEGAGTA_CACHE_WINDOW": "-2", "MEGAGTA_CACHE_COST_RATE": "-1"},
                {"MEGAGTA_CACHE_WINDOW": "0"}, {"MEGAGTA_CACHE_WINDOW": " +16"},      # (advisor r3: '' and values below -1 meant different modes on the two sides)
                {"MEGAGTA_CACHE_COST_KNEE": "65536", "MEGAGTA_CACHE_COST_RATE2": "16"}, {"MEGAGTA_CACHE_COST_KNEE": "0", "MEGAGTA_CACHE_COST_RATE2": "16"},
                {"MEGAGTA_CACHE_COST_KNEE": "4096", "MEGAGTA_CACHE_COST_RATE2": "1"}, {"MEGAGTA_CACHE_COST_RATE": "-2", "MEGAGTA_CACHE_COST_KNEE": "4096", "MEGAGTA_CACHE_COST_RATE2": "8"}):
        for key in ("MEGAGTA_CACHE_WINDOW", "MEGAGTA_CACHE_COST_RATE", "MEGAGTA_CACHE_COST_KNEE", "MEGAGTA_CACHE_COST_RATE2"):
            monkeypatch.delenv(key, raising=False)
        for key, v in env.items():
            monkeypatch.setenv(key, v)
        out = subprocess.run([exe, "searchplan"] + [str(n) for n in ns], capture_output=True, text=True, check=True, env=dict(os.environ)).stdout.split("\n")
        got = [tuple(int(x) for x in line.split()[1:]) for line in out if line.strip()]
        assert got == [search_dist.search_plan(n) for n in ns], env
    for bad in ("x", "8k", "4 ", "1.5"):                                   # not an integer: refused on both sides, never read as 0
        monkeypatch.setenv("MEGAGTA_CACHE_WINDOW", bad)
        r = subprocess.run([exe, "searchplan", "1000"], capture_output=True, text=True, env=dict(os.environ))
        assert r.returncode != 0 and "MEGAGTA_CACHE_WINDOW must be an integer" in r.stderr, bad
        with pytest.raises(SystemExit, match="MEGAGTA_CACHE_WINDOW must be an integer"):
            search_dist.window_and_rate(1000)
    for key in ("MEGAGTA_CACHE_WINDOW", "MEGAGTA_CACHE_COST_RATE", "MEGAGTA_CACHE_COST_KNEE", "MEGAGTA_CACHE_COST_RATE2"):
        monkeypatch.delenv(key, raising=False)
    assert search_dist.search_plan(400_000)[:2] == (8192, 1) and search_dist.search_plan(100_000)[:2] == (4096, 2)
    monkeypatch.setenv("MEGAGTA_CACHE_COST_KNEE", "1000"); monkeypatch.setenv("MEGAGTA_CACHE_COST_RATE2", "32")
    assert search_dist.window_and_rate(100_000) == (4096, (2, 1000, 32))


def test_graph_checkpoint_waits_for_the_worker_to_finish_the_files(tmp_path, monkeypatch):
    """advisor r3: in the worker `buildgraph` replies while a thread still writes PREFIX.sdbg.*; the checkpoint that says "graph built" is
    written only after the worker's "sync" request has confirmed the files (the reference writes cp.txt after the step's files are
    complete: megagta.py:380-385,586), and a writer failure is the build's failure: no checkpoint, non-zero exit"""
    import importlib
    from megagta_amd import megagta as drv

    class FakeWorker:
        def __init__(self, sync_rc):
            self.sync_rc, self.requests = sync_rc, []

        def request(self, fields):
            self.requests.append(fields[0])
            return self.sync_rc if fields[0] == "sync" else 0

        def close(self):
            self.requests.append("close")

    for sync_rc in (0, 1):
        drv = importlib.reload(drv)
        out = tmp_path / f"run{sync_rc}"
        drv.opt.out_dir = str(out) + "/"
        drv.opt.temp_dir = drv.opt.out_dir + "tmp/"
        os.makedirs(drv.opt.temp_dir)
        drv.opt.k_list, drv.opt.lib = [29, 44], drv.opt.temp_dir + "reads.lib"
        w = FakeWorker(sync_rc)
        drv.worker = w
        monkeypatch.setattr(drv, "run_step", lambda cmd, what, stdin_path=None, stdout_path=None: w.requests.append(cmd[1]))
        cp_path = drv.opt.temp_dir + "cp.txt"
        drv.build_graph(29, "")
        assert not os.path.exists(cp_path) or open(cp_path).read() == ""          # the files may still be in flight: nothing is promised yet
        # (round 5: `denovo` in the worker replies while a thread still writes PREFIX.contigs.fa -- its checkpoint waits like the build's;
        # the first step whose files are complete when it returns asks for the "sync" and writes the three of them)
        drv.assemble(29)
        assert w.requests == ["buildgraph", "denovo"]
        assert not os.path.exists(cp_path) or open(cp_path).read() == ""
        drv.opt.gene_info = {"g": ("f", "r", "a")}
        drv.find_seed(29, "g")                                             # (its checkpoint queues behind theirs: cp.txt is an ordered log)
        assert w.requests == ["buildgraph", "denovo", "findstart"]
        assert not os.path.exists(cp_path) or open(cp_path).read() == ""
        if sync_rc == 0:
            drv.flush_deferred_cp()                                       # what the first step with a checkpoint of its own (the search's filters) and the end of the run do
            assert w.requests == ["buildgraph", "denovo", "findstart", "sync"]
            assert open(cp_path).read() == "0\tdone\n1\tdone\n2\tdone\n"
        else:
            with pytest.raises(SystemExit) as e:
                drv.flush_deferred_cp()
            assert e.value.code == 1 and "sync" in w.requests
            assert not os.path.exists(cp_path) or open(cp_path).read() == ""      # --continue re-builds the graph and the contigs
    # one process per step (no worker): the step returns when its files are complete, the checkpoint follows at once
    drv = importlib.reload(drv)
    out = tmp_path / "run_steps"
    drv.opt.out_dir = str(out) + "/"
    drv.opt.temp_dir = drv.opt.out_dir + "tmp/"
    os.makedirs(drv.opt.temp_dir)
    drv.opt.k_list, drv.opt.lib = [29, 44], drv.opt.temp_dir + "reads.lib"
    monkeypatch.setattr(drv, "run_step", lambda *a, **k: None)
    drv.build_graph(29, "")
    assert open(drv.opt.temp_dir + "cp.txt").read() == "0\tdone\n"


def test_bucket_range_of_the_count_scan_without_a_division():
    """the count scan of a multi-range build finds the range of an item's bucket as ((b - b_lo) * ceil(2^32 / width)) >> 32 (sdbg_build.hip,
    ScanArgs::multi_magic) instead of dividing per item: exact for every width and every offset a 65536-bucket space can produce"""
    x = np.arange(0, 65536, dtype=np.uint64)
    for d in range(1, 65537):
        m = np.uint64(((1 << 32) + d - 1) // d)
        assert np.array_equal((x * m) >> np.uint64(32), x // np.uint64(d)), d
