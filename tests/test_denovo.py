"""`denovo` (assembler.cpp:98-167): tips, bubbles, contigs of an intermediate k.

The reference's loops race between threads (its 8-thread output differs from its 1-thread output on the same graph), so the golden
vectors are its ONE-thread run (tests/golden/make_golden_denovo.py).  CPU: the oracle's restatement against them, byte for byte.
GPU: the device path (mgta_denovo through the C ABI) against the goldens and, on seeded strain mixes, against the oracle."""
import ctypes as C
import gzip
import json
import os

import numpy as np
import pytest

from megagta_amd import readlib, synth

CODE = {c: i for i, c in enumerate("ACGT")}


def _golden(golden_dir):
    with open(os.path.join(golden_dir, "denovo", "expected.json")) as f:
        return json.load(f)


def _reads(golden_dir, name):
    with gzip.open(os.path.join(golden_dir, "denovo", name + ".fa.gz"), "rt") as f:
        return [np.array([CODE[c] for c in l.strip()], dtype=np.uint8) for l in f if not l.startswith(">")]


def _oracle_stream(oracle, reads, k, min_count):
    packed, start = readlib.pack_for_build(reads)
    if min_count == 1:
        return oracle.Stream.build(packed, start, k, threads=4)
    return oracle.Stream.build_solid(packed, start, k, min_count, False, threads=4)


CASES = ["strains_k29", "errors_k31", "tricky_k21", "tricky_k44"]


@pytest.mark.parametrize("name", CASES)
def test_oracle_vs_reference_one_thread(golden_dir, oracle, name):
    """tips at every doubling length, tied bubbles (the last branch wins: a different allele per strand), Pop undoing itself on shared inner
    edges, hairpins, tandem repeats, paths skipped because the other strand's walk locked them, --no_bubble, tips kept, min_contig"""
    case = _golden(golden_dir)[name]
    st = _oracle_stream(oracle, _reads(golden_dir, name), case["k"], case["min_count"])
    for run in case["runs"]:
        text, stats = oracle.Graph(st).denovo(run["max_tip_len"], run["no_bubble"], run["min_contig"])
        assert text == run["contigs"], (name, run["max_tip_len"], run["no_bubble"])
        assert f"{stats['n_contigs']} {stats['total_len']}\n" == run["info"]
        assert stats["n_contigs"] > 10


def test_c_abi_exports_denovo():
    from megagta_amd import _lib
    L = _lib.load()
    assert L.mgta_denovo and L.mgta_host_free
    out, n = C.c_void_p(), C.c_uint64()
    assert L.mgta_denovo(None, 150, 0, 0, C.byref(out), C.byref(n), None) == -1        # MGTA_EINVAL, no device touched


# ---- device ------------------------------------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def ctx():
    from megagta_amd import api
    return api.Context(0)


@pytest.mark.gpu
@pytest.mark.parametrize("name", CASES)
def test_device_vs_golden(golden_dir, ctx, name):
    """reads -> device build -> resident graph -> device denovo == the reference binary's one-thread files"""
    from megagta_amd import api
    case = _golden(golden_dir)[name]
    packed, start = readlib.pack_for_build(_reads(golden_dir, name))
    for run in case["runs"]:
        ctx.build_sdbg(ctx.upload_reads(packed, start), case["k"], min_count=case["min_count"], collect=False)
        g = api.Graph(ctx, None, case["k"])
        text, stats = g.denovo(run["max_tip_len"], run["no_bubble"], run["min_contig"])
        assert text == run["contigs"], (name, run["max_tip_len"], run["no_bubble"])
        assert f"{stats['n_contigs']} {stats['total_len']}\n" == run["info"]
        g.free()


@pytest.mark.gpu
@pytest.mark.parametrize("seed", range(8))
def test_device_vs_oracle_seeded(ctx, oracle, seed):
    """strain mixes of any k / min_count / variant density (dense SNPs give overlapping bubble candidates: several ordered rounds)"""
    from megagta_amd import api
    rng = np.random.default_rng(500 + seed)
    k = int(rng.choice([15, 21, 29, 31, 32, 44, 63, 64, 95]))
    mc = int(rng.choice([1, 2]))
    reads = synth.make_strain_mix(900 + seed, n_genomes=int(rng.integers(2, 6)), genome_len=int(rng.integers(1500, 4000)), read_len=int(max(100, k + 40)),
                                  snp_every=int(rng.choice([20, 35, 60, 150])), tricky=bool(seed & 1))
    st = _oracle_stream(oracle, reads, k, mc)
    for opts in ((150, False, k + 2), (-1, False, 0), (150, True, 0), (0, False, 0), (7, False, k + 10)):
        want, wst = oracle.Graph(st).denovo(*opts)
        got, gst = api.Graph(ctx, st.edges()).denovo(*opts)
        assert got == want, (seed, k, mc, opts)
        assert (gst["n_tips"], gst["n_bubbles"], gst["n_contigs"], gst["total_len"]) == (wst["n_tips"], wst["n_bubbles"], wst["n_contigs"], wst["total_len"])


@pytest.mark.gpu
def test_device_ordered_rounds(ctx, oracle):
    """variants closer than k: bubble candidates whose searches read each other's edges, so the window is cut at the first one that lost a
    stamp and popping takes several ordered rounds; the result is still the sequential loop's"""
    from megagta_amd import api
    reads = synth.make_strain_mix(77, n_genomes=6, genome_len=4000, snp_every=18, cov=30)
    st = _oracle_stream(oracle, reads, 21, 2)
    want, wst = oracle.Graph(st).denovo(150, False, 0)
    got, gst = api.Graph(ctx, st.edges()).denovo(150, False, 0)
    assert got == want and gst["n_bubbles"] == wst["n_bubbles"] > 50
    assert gst["n_bubble_rounds"] >= 2 and gst["n_bubble_candidates"] >= gst["n_bubbles"]
    # tiny windows (pending candidates carried from window to window) and a reach limit that no region fits (every candidate holds back
    # all higher ones: one commit per round at worst) must give the same contigs
    # ... and so must a stamp table (the hash of (edge -> round, rank) that replaced round 2's 8 bytes per edge) of 256 slots: most
    # candidates cannot stamp their reach, they and everything above them wait, the next window is a quarter of the size
    for knob, value in (("MGTA_DENOVO_WINDOW", "64"), ("MGTA_DENOVO_REACH_MAX", "8"), ("MGTA_DENOVO_STAMP_LOG2", "8"), ("MGTA_DENOVO_STAMP_LOG2", "11")):
        os.environ[knob] = value
        try:
            got2, gst2 = api.Graph(ctx, st.edges()).denovo(150, False, 0)
        finally:
            del os.environ[knob]
        assert got2 == want and gst2["n_bubbles"] == wst["n_bubbles"] and gst2["n_bubble_rounds"] > gst["n_bubble_rounds"], (knob, value)
    # the reach walk takes eight lanes per candidate up to 512 edges and a whole wave beyond (round 3): which of the two walks a region
    # takes changes nothing, not even the number of rounds
    for value in ("2", "24"):
        os.environ["MGTA_DENOVO_NARROW_MAX"] = value
        try:
            got3, gst3 = api.Graph(ctx, st.edges()).denovo(150, False, 0)
        finally:
            del os.environ["MGTA_DENOVO_NARROW_MAX"]
        assert got3 == want and gst3["n_bubbles"] == wst["n_bubbles"] and gst3["n_bubble_rounds"] == gst["n_bubble_rounds"], value


@pytest.mark.gpu
def test_device_edge_cases(ctx, oracle):
    """a graph with nothing to clean, reads shorter than k (empty graph), and a second call on a consumed graph"""
    from megagta_amd import api
    rng = np.random.default_rng(3)
    genome = rng.integers(0, 4, 700).astype(np.uint8)
    reads = [genome[p:p + 120].copy() for p in range(0, 580, 7)]
    packed, start = readlib.pack_for_build(reads)
    st = oracle.Stream.build(packed, start, 31, threads=2)
    want, wst = oracle.Graph(st).denovo(150, False, 0)
    g = api.Graph(ctx, st.edges())
    got, gst = g.denovo(150, False, 0)
    assert got == want and gst["n_bubbles"] == 0 and gst["n_contigs"] == wst["n_contigs"] >= 1
    again, _ = g.denovo(150, False, 0)            # validity bits already consumed: same contigs again, nothing more to remove
    assert again == want
    short = [genome[:20].copy(), genome[30:50].copy()]
    packed, start = readlib.pack_for_build(short)
    est = oracle.Stream.build(packed, start, 31, threads=1)
    text, s = api.Graph(ctx, est.edges()).denovo(150, False, 0)
    assert text == "" and s["n_contigs"] == 0


@pytest.mark.gpu
@pytest.mark.parametrize("no_bubble", [True, False])
def test_device_validity_bits_after_cleaning(ctx, oracle, no_bubble):
    """not only the contigs: the graph itself (which edges tips and bubbles removed) is the oracle's, bit for bit"""
    from megagta_amd import api
    reads = synth.make_strain_mix(321, n_genomes=4, genome_len=3000, snp_every=45, tricky=True)
    st = _oracle_stream(oracle, reads, 31, 2)
    og, g = oracle.Graph(st), api.Graph(ctx, st.edges())
    assert np.array_equal(og.invalid_now(), g.invalid_bits())                       # tips and $ edges after the load
    before = int(np.unpackbits(g.invalid_bits().view(np.uint8)).sum())
    og.denovo(150, no_bubble, 0)
    g.denovo(150, no_bubble, 0)
    a, b = og.invalid_now(), g.invalid_bits()
    assert np.array_equal(a, b)
    assert int(np.unpackbits(b.view(np.uint8)).sum()) > before + 100
