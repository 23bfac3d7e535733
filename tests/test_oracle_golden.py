"""Pins the CPU oracle against golden vectors captured from the COMPILED REFERENCE
(tests/golden/make_golden.py; reference entry points cited there).  CPU-only.
"""
import hashlib
import os

import numpy as np
import pytest

from megagta_amd import readlib
from tests import helpers as H


def _stream_matches(e, fx):
    assert e.k == fx["k"] and e.words_per_tip == fx["words_per_tip"]
    assert int(e.records.size) == fx["num_edges"]
    assert int(e.large.size) == fx["num_large"]
    assert hashlib.md5(e.bucket_items.astype("<i8").tobytes()).hexdigest() == fx["bucket_md5"]
    assert [int(x) for x in e.records[:256]] == fx["head_records"]
    assert [int(x) for x in e.large[:64]] == fx["large"]
    assert [int(x) for x in e.tips[: 8 * e.words_per_tip]] == fx["head_tips"]
    assert e.md5() == fx["md5"]


@pytest.mark.parametrize("k", [29, 35, 44])
def test_sdbg_stream_toy(oracle, golden_dir, k):
    """Edge stream of the oracle build == the reference's buildgraph output (cx1_read2sdbg_s2.cpp:742-835)."""
    packed, start = readlib.load_for_build(os.path.join(golden_dir, "toy", "reads.lib"))
    e = oracle.Stream.build(packed, start, k, threads=4).edges()
    _stream_matches(e, H.load_streams(os.path.join(golden_dir, "toy", "sdbg_streams.json"))[str(k)])
    # per read 2(L-k)+4 sort items when every position is solid (SURVEY.md §8 size table); no palindromes for odd k+1
    if (k + 1) % 2 == 1:
        assert e.n_items_sorted == 6000 * (2 * (150 - k) + 4)


@pytest.mark.parametrize("k", [21, 29, 31, 44, 47, 63])
def test_sdbg_stream_ragged(oracle, golden_dir, k):
    """Ragged lengths, reads shorter than k+1, N->G, palindromic (k+1)-mers, multiplicity > 254."""
    packed, start = readlib.load_for_build(os.path.join(golden_dir, "ragged", "reads.lib"))
    e = oracle.Stream.build(packed, start, k, threads=2).edges()
    fx = H.load_streams(os.path.join(golden_dir, "ragged", "sdbg_streams.json"))[str(k)]
    _stream_matches(e, fx)
    if k <= 47:
        assert fx["num_large"] > 0      # the >254 path is really exercised


def _counting_md5(hist) -> str:
    """PREFIX.counting as s1_post_proc writes it (cx1_read2sdbg_s1.cpp:923-930): `i cumulative_count` for i = 1..65535"""
    acc = np.cumsum(hist[1:])
    return hashlib.md5("".join(f"{i} {int(a)}\n" for i, a in zip(range(1, 65536), acc)).encode()).hexdigest()


@pytest.mark.parametrize("sub", ["toy", "ragged"])
def test_sdbg_stream_min_count_and_mercy(oracle, golden_dir, sub):
    """Stage 1 of the oracle (solid (k+1)-mers, mercy edges, .counting) + stage 2 over the solid runs == the reference's
    `buildgraph -m M [--need_mercy]` (cx1_read2sdbg_s1.cpp, cx1_read2sdbg_s2.cpp:106-250)."""
    import json
    packed, start = readlib.load_for_build(os.path.join(golden_dir, sub, "reads.lib"))
    cases = json.load(open(os.path.join(golden_dir, sub, "sdbg_streams_solid.json")))
    ran = 0
    for tag, fx in cases.items():
        if fx.get("reference_crashed"):
            continue
        k, m, mercy = int(tag.split("_")[0][1:]), int(tag.split("_")[1][1:]), tag.endswith("_mercy")
        st = oracle.Stream.build_solid(packed, start, k, m, mercy, threads=4)
        _stream_matches(st.edges(), fx)
        assert _counting_md5(st.counting) == fx["counting_md5"], tag
        ran += 1
    assert ran >= 4


def _check_graph(oracle, stream, lines):
    g = oracle.Graph(stream)
    hdr, qs = H.parse_probe_graph(lines)
    assert g.size == int(hdr["size"][0]) and g.k == int(hdr["k"][0])
    assert [int(x) for x in hdr["f"]] == list(g.f)
    bv = g.bitvectors()
    for name, key in (("w", "fnv_w"), ("last", "fnv_last"), ("tip", "fnv_tip"), ("invalid", "fnv_invalid"),
                      ("multi1", "fnv_multi1"), ("tip_labels", "fnv_tiplabels")):
        assert H.fnv1a(bv[name]) == hdr[key][0], name
    for q in qs:
        e = q["e"]
        n, out = g.outgoing(e)
        assert (n, out) == (q["od"], q["out"])
        assert g.rank_last(e) == q["rank_last"]
        assert [g.rank_w(c, e) for c in range(9)] == q["rank_w"]
        assert g.select_last(q["rank_last"] - 1) == q["select_last"]
        if "fwd" in q:
            assert g.forward(e) == q["fwd"]
            assert g.label(e) == q["label"]
            assert g.incoming(e) == (q["id"], q["in"])
    return g


def test_graph_navigation_toy(oracle, golden_dir):
    """LoadFromMultiFile bit-vectors + Rank/Select/Forward/OutgoingEdges/IncomingEdges/Label answers."""
    packed, start = readlib.load_for_build(os.path.join(golden_dir, "toy", "reads.lib"))
    s = oracle.Stream.build(packed, start, 44, threads=4)
    _check_graph(oracle, s, H.gz_lines(os.path.join(golden_dir, "toy", "graph_k44.txt.gz")))


@pytest.mark.parametrize("k", [29, 47])
def test_graph_navigation_ragged(oracle, golden_dir, k):
    packed, start = readlib.load_for_build(os.path.join(golden_dir, "ragged", "reads.lib"))
    s = oracle.Stream.build(packed, start, k, threads=2)
    _check_graph(oracle, s, H.gz_lines(os.path.join(golden_dir, "ragged", f"graph_k{k}.txt.gz")))


def test_index_binary_search_edge(oracle, golden_dir):
    packed, start = readlib.load_for_build(os.path.join(golden_dir, "toy", "reads.lib"))
    g = oracle.Graph(oracle.Stream.build(packed, start, 44, threads=4))
    hits = 0
    for l in H.gz_lines(os.path.join(golden_dir, "toy", "index_k44.txt.gz")):
        kmer, ans = l.split()
        assert g.index_edge(kmer) == int(ans)
        hits += int(ans) >= 0
    assert hits > 50


@pytest.mark.parametrize("tag", ["for", "rev"])
def test_hmm_tables(oracle, golden_dir, tag):
    """Parser::readHMM tables + MostProbablePath heuristic, bit-for-bit (hex doubles)."""
    hm = oracle.Hmm(os.path.join(golden_dir, "toy", f"{tag}_enone.hmm"))
    ref = H.parse_probe_hmm(H.gz_lines(os.path.join(golden_dir, "toy", f"hmm_{tag}.txt.gz")))
    assert (hm.M, hm.A) == (ref["M"], ref["A"])
    assert list(hm.alpha) == ref["alpha"]
    for k in range(hm.M + 1):
        if k > 0:
            assert np.array_equal(hm.msc[k], ref["msc"][k])
        assert np.array_equal(hm.isc[k], ref["isc"][k])
        assert np.array_equal(hm.tsc[:, k], ref["tsc"][k])
        assert hm.maxm[k] == ref["maxm"][k]
        assert np.array_equal(hm.hcost[:, k], ref["h"][k])


def test_codon_tables(golden_dir):
    from megagta_amd import synth
    lines = dict(l.split() for l in H.gz_lines(os.path.join(golden_dir, "toy", "codon.txt.gz")))
    assert lines["fwd"] == synth._CODON_AA == lines["libseq"]
    rc = "".join(synth._CODON_AA[(3 - (i & 3)) * 16 + (3 - ((i >> 2) & 3)) * 4 + (3 - (i >> 4))] for i in range(64))
    assert lines["rc"] == rc


def _side_matches(res, d):
    assert res.ok == d["ok"]
    if d["ok"]:
        assert res.real_score == d["real"] and res.score == d["score"]     # bit-exact fp64
        assert (res.fval, res.length, res.state_no, chr(res.state), res.node_id) == \
               (d["fval"], d["length"], d["state_no"], d["state"], d["node"])
    assert res.n_closed == d["closed"]


@pytest.mark.parametrize("mode,fname,prune", [("cold", "astar_cold.txt.gz", 20), ("warm", "astar_warm.txt.gz", 20),
                                               ("cold", "astar_cold_prune0.txt.gz", 0)])
def test_astar(oracle, golden_dir, mode, fname, prune):
    """Per-seed A* results (both directions) == HMMGraphSearch::search on the reference, cold and warm term_nodes cache."""
    toy = os.path.join(golden_dir, "toy")
    packed, start = readlib.load_for_build(os.path.join(toy, "reads.lib"))
    g = oracle.Graph(oracle.Stream.build(packed, start, 44, threads=4))
    fw, rv = oracle.Hmm(os.path.join(toy, "for_enone.hmm")), oracle.Hmm(os.path.join(toy, "rev_enone.hmm"))
    S = oracle.Searcher(g, fw, rv, prune, 0.5)
    gold = H.parse_probe_astar(H.gz_lines(os.path.join(toy, fname)))
    assert len(gold) >= 12
    contigs = []
    for rec in gold:
        contig, R, L = S.search(rec["kmer"], rec["start_state"], cold=(mode == "cold"))
        _side_matches(R, rec["R"])
        _side_matches(L, rec["L"])
        assert contig == rec["contig"]
        contigs.append(contig)
    if mode == "warm" and prune == 20:
        # and the reference binary's own output file (search ... 1 thread), hmm_graph_search.h:79
        fasta = H.gz_lines(os.path.join(toy, "44_raw_contigs_rplB.fasta.gz"))
        assert fasta[0::2] == [f">rplB_contig_{2 * i}_contig_{2 * i + 1}" for i in range(len(gold))]
        assert fasta[1::2] == contigs


@pytest.mark.parametrize("case,mode", [("m600", "cold"), ("m600", "warm"), ("m1200", "warm")])
def test_astar_models_beyond_the_lds(oracle, golden_dir, tmp_path, case, mode):
    """600- and 1200-column models (tests/golden/bigm, from the reference's probe): the oracle's searcher on the models the device reads
    from global memory instead of LDS -- per seed, both directions, scores bit-equal"""
    packed, start, gdir, cold, warm = H.bigm_case(golden_dir, case, str(tmp_path))
    g = oracle.Graph(oracle.Stream.build(packed, start, 44, threads=4))
    S = oracle.Searcher(g, oracle.Hmm(os.path.join(gdir, "for_enone.hmm")), oracle.Hmm(os.path.join(gdir, "rev_enone.hmm")), 20, 0.5)
    gold = cold if mode == "cold" else warm
    assert len(gold) >= 30
    for rec in gold:
        contig, R, L = S.search(rec["kmer"], rec["start_state"], cold=(mode == "cold"))
        _side_matches(R, rec["R"])
        _side_matches(L, rec["L"])
        assert contig == rec["contig"]


@pytest.mark.parametrize("case,k,threads", [("toy", 44, 4), ("toy", 29, 2), ("ragged", 47, 3), ("ragged", 21, 8)])
def test_oracle_sdbg_reader_pinned_by_fresh_reference_files(oracle, golden_dir, tmp_path, case, k, threads):
    """the golden stream MD5s are computed by the oracle's own `.sdbg` reader: pin that reader directly.  A raw-file MD5 cannot serve
    (the reference spreads the buckets over its `.sdbg.N` files by thread timing: two runs of the same command write different bytes),
    so the reference binary is run HERE, with another thread count than the generator used, and the oracle's decoding of the files it
    has just written must be the stream the fixture describes; the decoded file set is checked to hold every bucket exactly once."""
    import glob
    import subprocess
    ref = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "_ref", "megagta")
    if not os.path.exists(ref):
        pytest.skip("oracle/_ref/megagta (the prebuilt reference) is not present")
    fx = H.load_streams(os.path.join(golden_dir, case, "sdbg_streams.json"))[str(k)]
    prefix = str(tmp_path / "g")
    subprocess.run([ref, "buildgraph", "-k", str(k), "-m", "1", "--host_mem", "2000000000", "--mem_flag", "1", "--gpu_mem", "0", "--output_prefix",
                    prefix, "--num_cpu_threads", str(threads), "--num_output_threads", "1", "--read_lib_file", os.path.join(golden_dir, case, "reads.lib")],
                   check=True, capture_output=True)
    assert len(glob.glob(prefix + ".sdbg.*")) >= 1
    e = oracle.Stream.read(prefix).edges()
    _stream_matches(e, fx)
    assert int(e.bucket_items.sum()) == fx["num_edges"]
