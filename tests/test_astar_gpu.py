"""Batched HMM-guided A* on the GPU vs the reference's per-seed results (golden, cold cache) and the oracle.
Bar: contig strings exact; path log-probabilities (real_score, score) within 1e-4 relative -- in fact bit-equal;
closed-node counts equal (path-identical search)."""
import os

import numpy as np
import pytest

from megagta_amd import hmm as hmmlib
from megagta_amd import readlib, synth
from tests import helpers as H

pytestmark = pytest.mark.gpu
REL = 1e-4   # tolerance named by BASELINE.json north_star for HMM path log-probabilities


@pytest.fixture(scope="module")
def ctx():
    from megagta_amd import api
    c = api.Context(0)
    yield c
    c.close()


@pytest.fixture(scope="module")
def toy(ctx, golden_dir):
    from megagta_amd import api
    d = os.path.join(golden_dir, "toy")
    packed, start = readlib.load_for_build(os.path.join(d, "reads.lib"))
    stream = ctx.build_sdbg(ctx.upload_reads(packed, start), 44)
    g = api.Graph(ctx, stream)
    fw = api.DeviceHmm(ctx, hmmlib.parse_hmm(os.path.join(d, "for_enone.hmm")))
    rv = api.DeviceHmm(ctx, hmmlib.parse_hmm(os.path.join(d, "rev_enone.hmm")))
    return g, fw, rv, d


class _Meta20k:
    """the 20 k-read metagenome of five tests below, made ONCE per module: reads, the device graph, the two models, 400 synthetic seeds
    (the first 300 are test_vs_oracle_bigger_graph's) and -- built when first asked for -- the oracle's graph of the same reads"""

    def __init__(self, ctx, tmpdir):
        from megagta_amd import api
        mg = synth.make_metagenome(20000, 150, (("rplB", 120),), seed=9, reads_per_genome=1000)
        self.mg = mg
        self.packed, self.start = synth.pack_reads_for_build(mg.reads)
        stream = ctx.build_sdbg(ctx.upload_reads(self.packed, self.start), 44)
        synth.write_gene_models(mg.genes, tmpdir)
        self.fpath, self.rpath = os.path.join(tmpdir, "rplB", "for_enone.hmm"), os.path.join(tmpdir, "rplB", "rev_enone.hmm")
        self.seeds = synth.synthetic_seeds(mg.genes[0], 45, 400, seed=4)
        self.g = api.Graph(ctx, stream)
        self.fw, self.rv = api.DeviceHmm(ctx, hmmlib.parse_hmm(self.fpath)), api.DeviceHmm(ctx, hmmlib.parse_hmm(self.rpath))
        self.kmers, self.states = [s[0] for s in self.seeds], [s[1] - 1 for s in self.seeds]
        self._og = None

    def oracle_graph(self, oracle):
        if self._og is None:
            self._og = oracle.Graph(oracle.Stream.build(self.packed, self.start, 44, threads=8))
        return self._og


@pytest.fixture(scope="module")
def meta20k(ctx, tmp_path_factory):
    return _Meta20k(ctx, str(tmp_path_factory.mktemp("meta20k")))


def request_id(mode):
    return {(16, 0): "g16", (64, 0): "g64", (16, 7): "g16-grow", (64, 7): "g64-grow", (8, 0): "g8", (8, 7): "g8-grow"}[tuple(mode)]


def _close(a, b):
    return a == b or abs(a - b) <= REL * max(abs(a), abs(b))


def _check_side(got, ref):
    assert got["ok"] == ref["ok"]
    if ref["ok"]:
        assert _close(got["real_score"], ref["real"]) and _close(got["score"], ref["score"])
        assert got["real_score"] == ref["real"] and got["score"] == ref["score"]        # bit-equal fp64 in practice
        assert (got["fval"], got["length"], got["state_no"], got["state"], got["node_id"]) == \
               (ref["fval"], ref["length"], ref["state_no"], ref["state"], ref["node"])
    assert got["n_closed"] == ref["closed"]


@pytest.fixture(params=[(16, 0), (64, 0), (16, 7), (64, 7), (8, 0), (8, 7)], ids=["g16", "g64", "g16-grow", "g64-grow", "g8", "g8-grow"])
def search_mode(request, ctx):
    """lanes per search (8: eight searches per wavefront, the (first edge, second edge) pairs walked in two passes; 16: four searches;
    64: one) x base arena (0 = default 8192 nodes; 7 = 128 nodes: every search
    of the goldens then outgrows its base arena and re-hashes several times)"""
    group, log_b0 = request.param
    os.environ["MGTA_ASTAR_GROUP"] = str(group)
    ctx.set_search_arena(log_b0, 0)
    yield request.param
    os.environ.pop("MGTA_ASTAR_GROUP", None)
    ctx.set_search_arena(0, 0)


@pytest.mark.parametrize("fname,prune", [("astar_cold.txt.gz", 20), ("astar_cold_prune0.txt.gz", 0)])
def test_cold_cache_vs_reference(toy, fname, prune, search_mode):
    from megagta_amd import api
    g, fw, rv, d = toy
    gold = H.parse_probe_astar(H.gz_lines(os.path.join(d, fname)))
    res, st = api.astar_search(g, fw, rv, [r["kmer"] for r in gold], [r["start_state"] for r in gold], prune, 0.5)
    assert st["hmm_in_lds"] == 1
    if search_mode[1]:
        assert st["n_grown"] > 0 and st["n_rehash"] > 0 and st["n_retries"] == 0
    for r, ref in zip(res, gold):
        _check_side(r.right_side, ref["R"])
        _check_side(r.left_side, ref["L"])
        assert r.contig(ref["kmer"]) == ref["contig"]
    assert st["n_expansions"] == sum(r["R"]["closed"] + r["L"]["closed"] for r in gold) + \
           sum(int(x.right_side["n_expanded"] > x.right_side["n_closed"]) + int(x.left_side["n_expanded"] > x.left_side["n_closed"]) for x in res)


@pytest.mark.parametrize("case", ["m600", "m1200"])
def test_models_beyond_the_lds_vs_reference(ctx, golden_dir, tmp_path, case, search_mode):
    """astar_kernel<G, false>: a 600- / 1200-column model's tables ((M + 1)(A + 11) * 8 B = 149 / 298 KB) do not fit a CU's 160 KB of LDS
    beside the heap tops, the kernel reads them from device memory (the reference has no bound on M: profile_hmm.h:11-100).  Goldens from
    the reference's probe (tests/golden/bigm): cold per seed in every lane mode, and the sequential shared-cache run (window 1 ==
    `search ... 1`)"""
    from megagta_amd import api
    # (every lane mode on one of the two models, not 12 combinations of ~25 s: the suite has a time budget)
    # (("m1200", "g64") = astar_kernel<64, false>: one search per wavefront with the tables in device memory, cold only)
    if (case, request_id(search_mode)) not in {("m600", "g16"), ("m600", "g8-grow"), ("m600", "g64"), ("m1200", "g16-grow"), ("m1200", "g8"), ("m1200", "g64")}:
        pytest.skip("combination left to the other model")
    packed, start, gdir, cold, warm = H.bigm_case(golden_dir, case, str(tmp_path))
    g = api.Graph(ctx, ctx.build_sdbg(ctx.upload_reads(packed, start), 44))
    fw = api.DeviceHmm(ctx, hmmlib.parse_hmm(os.path.join(gdir, "for_enone.hmm")))
    rv = api.DeviceHmm(ctx, hmmlib.parse_hmm(os.path.join(gdir, "rev_enone.hmm")))
    # (the sequential run once per table variant and model: one search at a time per direction takes half a minute for 1200 columns)
    for gold, mode in ((cold, 0), (warm, 1))[: 2 if (case, request_id(search_mode)) in {("m1200", "g8"), ("m600", "g8-grow"), ("m600", "g64")} else 1]:
        res, st = api.astar_search(g, fw, rv, [r["kmer"] for r in gold], [r["start_state"] for r in gold], 20, 0.5, cache_mode=mode)
        # the variant under test is the one that ran (one search per wavefront keeps so little of the heap in LDS that 600 columns still fit)
        assert st["hmm_in_lds"] == (1 if (search_mode[0] == 64 and case == "m600") else 0)
        for r, ref in zip(res, gold):
            _check_side(r.right_side, ref["R"])
            _check_side(r.left_side, ref["L"])
            assert r.contig(ref["kmer"]) == ref["contig"]
    assert st["max_search_expansions"] > 10000                        # (these searches are long ones: ~10^5 expansions per seed)


def test_giving_up_the_order_is_opt_in_and_says_so(ctx, oracle, monkeypatch, meta20k):
    """advisor r4: a batch whose searches in flight outgrow their pool used to give up the ORDER of its cache sharing on its own (timing-
    dependent contigs by default on large inputs, and a branch no test ran).  Now the order is held unless MEGAGTA_SEARCH_ALLOW_UNORDERED=1:
    the same starved pool gives the roomy run's result without it, and with it the batch says `order_abandoned`, every search still ends,
    and every contig is a path of the graph through its seed k-mer (which of the admissible paths: a matter of timing, as in the reference's
    multi-thread search, search.cpp:182-189)"""
    from megagta_amd import api
    g, fw, rv, kmers, states, packed, start = meta20k.g, meta20k.fw, meta20k.rv, meta20k.kmers, meta20k.states, meta20k.packed, meta20k.start
    if True:
        monkeypatch.delenv("MEGAGTA_SEARCH_ALLOW_UNORDERED", raising=False)
        want, st0 = api.astar_search(g, fw, rv, kmers, states, 0, 0.5, cache_mode=8)          # roomy, ordered; prune 0: the largest searches
        assert st0["order_abandoned"] == 0
        try:
            ctx.set_search_arena(7, 12288 << 10)                                             # 128-node base arenas, 12 MB for 800 searches
            held, st1 = api.astar_search(g, fw, rv, kmers, states, 0, 0.5, cache_mode=8)
            monkeypatch.setenv("MEGAGTA_SEARCH_ALLOW_UNORDERED", "1")
            free, st2 = api.astar_search(g, fw, rv, kmers, states, 0, 0.5, cache_mode=8)
        finally:
            ctx.set_search_arena(0, 0)
        # default: the order is held whatever it costs -- the roomy run's contigs, scores and counts
        assert st1["order_abandoned"] == 0 and st1["n_expansions"] == st0["n_expansions"]
        for a, b, km in zip(held, want, kmers):
            assert a.contig(km) == b.contig(km) and a.right_side == b.right_side and a.left_side == b.left_side
        # opted in: the batch gave the order up (thousands of refused requests) and said so; every seed has its contig, every contig is a walk
        # in the graph that contains its seed
        assert st2["order_abandoned"] == 1, st2
        og = meta20k.oracle_graph(oracle)
        n_checked = 0
        for r, km in zip(free, kmers):
            c = r.contig(km)
            assert km.lower() in c
            if og.index_edge(km.upper()) < 0:
                continue                                                 # (a synthetic seed that is not in the graph stays as it is)
            for i in range(0, len(c) - 44, 7):                           # every 7th (k+1)-mer of the contig is an edge of the graph
                assert og.index_edge(c[i:i + 45].upper()) >= 0, (km, i)
                n_checked += 1
        assert n_checked > 1000


def test_vs_oracle_bigger_graph(ctx, oracle, meta20k):
    """1 gene, 20k reads, synthetic seeds (incl. k-mers absent from the graph): GPU == oracle per seed"""
    from megagta_amd import api
    g, fw, rv, fpath, rpath = meta20k.g, meta20k.fw, meta20k.rv, meta20k.fpath, meta20k.rpath
    seeds = meta20k.seeds[:300]                                      # (synthetic_seeds(.., 300, seed=4) is a prefix of the 400)
    assert seeds == synth.synthetic_seeds(meta20k.mg.genes[0], 45, 300, seed=4)
    if True:
        res, st = api.astar_search(g, fw, rv, [s[0] for s in seeds], [s[1] - 1 for s in seeds], 20, 0.5)
        og = meta20k.oracle_graph(oracle)
        S = oracle.Searcher(og, oracle.Hmm(fpath), oracle.Hmm(rpath), 20, 0.5)
        nexp = 0
        for (kmer, pos), r in zip(seeds, res):
            contig, R, L = S.search(kmer, pos - 1, cold=True)
            assert r.contig(kmer) == contig
            for got, ref in ((r.right_side, R), (r.left_side, L)):
                assert got["ok"] == ref.ok and got["n_closed"] == ref.n_closed and got["n_expanded"] == ref.n_expanded
                assert got["partial"] == ref.partial and got["n_opened"] == ref.n_opened
                if ref.ok:
                    assert got["real_score"] == ref.real_score and got["score"] == ref.score and got["fval"] == ref.fval
                nexp += ref.n_expanded
        assert st["n_expansions"] == nexp and nexp > 10000


def test_longest_first_order_of_cold_batches_changes_no_result(ctx, meta20k, monkeypatch):
    """independent (cold) searches are started in descending order of the model columns their side has to cover (the long ones run beside
    the bulk, not after it): per seed the contigs, scores and counts are those of the run in seed order (MGTA_ASTAR_LPT=0)"""
    from megagta_amd import api
    monkeypatch.setenv("MGTA_ASTAR_LPT", "0")
    want, st0 = api.astar_search(meta20k.g, meta20k.fw, meta20k.rv, meta20k.kmers, meta20k.states, 20, 0.5)
    monkeypatch.delenv("MGTA_ASTAR_LPT")
    got, st1 = api.astar_search(meta20k.g, meta20k.fw, meta20k.rv, meta20k.kmers, meta20k.states, 20, 0.5)
    assert st0["n_expansions"] == st1["n_expansions"] > 10000
    # the moment the last seed was taken lies inside the launch (what follows is the tail: mgta_astar_stats.ms_queue_drained)
    for st in (st0, st1):
        assert 0.0 <= st["ms_queue_drained"] <= st["ms_kernel"] * 1.05 + 1.0, st
    for a, b, km in zip(got, want, meta20k.kmers):
        assert a.contig(km) == b.contig(km) and a.right_side == b.right_side and a.left_side == b.left_side


def test_random_line_probe_reports_rates_and_latencies(ctx):
    """mgta_probe_random_lines (the yardstick of the search leg's roofline, csrc/probe.hip) on a small table: every configuration reads the lines it
    says, independent reads are faster than a pointer chase, a chase with a store in every step is no faster than the bare chase, and bad
    configurations are refused"""
    from megagta_amd import api
    cfgs = [(4, 8, 4, 0, 512), (4, 8, 1, 1, 512), (4, 8, 1, 2, 512), (4, 8, 1, 3, 512), (1, 1, 1, 1, 2048)]
    res = ctx.probe_random_lines(256 << 20, cfgs)
    n_cu = res[0]["lines"] // (4 * 8 * 4 * 512)
    assert n_cu >= 8
    for (w, g, u, d, steps), r in zip(cfgs, res):
        assert r["lines"] == n_cu * w * g * u * steps and r["lines_in_flight_per_cu"] == w * g * u
        assert r["ms"] > 0 and r["gb_per_s"] > 0 and r["ns_per_step"] > 0
    assert res[0]["gb_per_s"] > 2 * res[1]["gb_per_s"]                  # 128 independent lines in flight per CU against 32 dependent chains
    assert res[4]["ns_per_step"] > 100                                  # one dependent line on an idle chip: hundreds of nanoseconds
    assert res[2]["ns_per_step"] > 0.9 * res[1]["ns_per_step"]
    for bad in ((0, 8, 1, 0, 16), (4, 9, 1, 0, 16), (4, 8, 3, 0, 16), (4, 8, 1, 4, 16), (4, 8, 1, 0, 0)):
        with pytest.raises(api.MegaGtaError):
            ctx.probe_random_lines(256 << 20, [bad])


def test_bad_seed_is_loud(toy):
    from megagta_amd import api
    g, fw, rv, d = toy
    with pytest.raises(api.MegaGtaError):
        api.astar_search(g, fw, rv, ["A" * 45], [95], 20, 0.5)      # model position + 15 codons > M = 100


def test_pool_exhaustion_reruns_only_the_searches_it_hit(toy, ctx):
    """a pool far too small for the searches in flight: the searches it cannot hold are run again (fewer at a time) and every result
    still equals the reference's; a pool that cannot hold even one search is a loud error, never a wrong or missing contig"""
    from megagta_amd import api
    g, fw, rv, d = toy
    gold = H.parse_probe_astar(H.gz_lines(os.path.join(d, "astar_cold_prune0.txt.gz")))
    kmers, states = [r["kmer"] for r in gold], [r["start_state"] for r in gold]
    seen_retry = False
    try:
        sizes = []
        for kb in (16, 32, 64, 128, 256, 512, 1024, 4096, 16384, 65536):
            ctx.set_search_arena(7, kb << 10)
            try:
                res, st = api.astar_search(g, fw, rv, kmers, states, 0, 0.5)
            except api.MegaGtaError as e:
                assert "do not fit" in str(e)
                sizes.append((kb, "error"))
                continue
            sizes.append((kb, st["n_retries"]))
            seen_retry |= st["n_retries"] > 0
            for r, ref in zip(res, gold):
                _check_side(r.right_side, ref["R"])
                _check_side(r.left_side, ref["L"])
                assert r.contig(ref["kmer"]) == ref["contig"]
            if st["n_retries"] == 0:
                break
    finally:
        ctx.set_search_arena(0, 0)
    assert seen_retry, sizes


def test_warm_sequential_equals_reference_search_1thread(toy, search_mode):
    """cache_mode 1: shared term_nodes caches with ordered commits == `megagta search ... 1` (per seed AND the FASTA file)"""
    from megagta_amd import api
    g, fw, rv, d = toy
    gold = H.parse_probe_astar(H.gz_lines(os.path.join(d, "astar_warm.txt.gz")))
    res, st = api.astar_search(g, fw, rv, [r["kmer"] for r in gold], [r["start_state"] for r in gold], 20, 0.5, cache_mode=1)
    for r, ref in zip(res, gold):
        _check_side(r.right_side, ref["R"])
        _check_side(r.left_side, ref["L"])
        assert r.contig(ref["kmer"]) == ref["contig"]
    fasta = H.gz_lines(os.path.join(d, "44_raw_contigs_rplB.fasta.gz"))
    assert fasta[1::2] == [r.contig(ref["kmer"]) for r, ref in zip(res, gold)]
    cold = H.parse_probe_astar(H.gz_lines(os.path.join(d, "astar_cold.txt.gz")))
    assert st["n_expansions"] < sum(c["R"]["closed"] + c["L"]["closed"] for c in cold) / 3      # the cache really short-cuts


@pytest.mark.parametrize("window,rate", [(1, 0), (4, 0), (64, 0), (1, 1), (4, 16), (16, 64), (64, 4), (8, -1), (32, -3),   # rate < 0: c * |rate| seeds
                                         (4, (1, 64, 16)), (16, (4, 500, 64)), (2, (2, 1, 2000))])                      # (rate, knee, rate beyond the knee)
def test_windowed_warm_vs_oracle(ctx, oracle, window, rate, meta20k):
    """cache_mode B (+ cost rate R): the path seed j found with c_j expansions is seen by the seeds >= j + B + c_j // R (R = 0: no cost term;
    (R, K, R2): c_j // R up to K expansions, K // R + (c_j - K) // R2 beyond -- mgta_ctx_set_search_cost_curve);
    deterministic whatever the GPU scheduling; == the oracle run sequentially with the same rule"""
    from megagta_amd import api
    g, fw, rv, fpath, rpath, seeds = meta20k.g, meta20k.fw, meta20k.rv, meta20k.fpath, meta20k.rpath, meta20k.seeds
    if True:
        runs = [api.astar_search(g, fw, rv, [s[0] for s in seeds], [s[1] - 1 for s in seeds], 20, 0.5, cache_mode=window, cost_rate=rate) for _ in range(2)]
        og = meta20k.oracle_graph(oracle)
        S = oracle.Searcher(og, oracle.Hmm(fpath), oracle.Hmm(rpath), 20, 0.5)
        S.clear_cache()
        S.set_window(window)
        S.set_cost_rate(rate)
        for i, (kmer, pos) in enumerate(seeds):
            contig, R, L = S.search(kmer, pos - 1, cold=False)
            for res, _ in runs:
                r = res[i]
                assert r.contig(kmer) == contig, (window, rate, i)
                for got, ref in ((r.right_side, R), (r.left_side, L)):
                    assert got["ok"] == ref.ok and got["n_closed"] == ref.n_closed and got["n_expanded"] == ref.n_expanded
                    if ref.ok:
                        assert got["real_score"] == ref.real_score and got["fval"] == ref.fval


def test_window_mode_grows_in_place_instead_of_failing(ctx, meta20k):
    """the CLI's default mode (ordered-commit window) with base arenas of 128 nodes: every search outgrows its arena many times over
    (round 1 returned MGTA_EOVERFLOW from this mode when a search passed 2^18 nodes); results == the same run with roomy arenas"""
    from megagta_amd import api
    g, fw, rv, kmers, states = meta20k.g, meta20k.fw, meta20k.rv, meta20k.kmers, meta20k.states
    if True:
        want, st0 = api.astar_search(g, fw, rv, kmers, states, 0, 0.5, cache_mode=8)          # prune 0: the largest searches
        try:
            ctx.set_search_arena(7, 0)
            got, st = api.astar_search(g, fw, rv, kmers, states, 0, 0.5, cache_mode=8)
        finally:
            ctx.set_search_arena(0, 0)
        # (every search outgrows its 128-node base arena: pages of nodes, of heap slots, and its hash table moves into a bucket of its own)
        assert st["n_grown"] > 300 and st["n_rehash"] >= st["n_grown"] // 2 and st["n_retries"] == 0 and st0["n_retries"] == 0
        assert st["n_expansions"] == st0["n_expansions"]
        for a, b, km in zip(got, want, kmers):
            assert a.contig(km) == b.contig(km) and a.right_side == b.right_side and a.left_side == b.left_side


def test_window_mode_with_a_starved_pool_is_still_the_roomy_result(ctx, meta20k):
    """ordered-commit window + a pool far too small for the searches in flight: a starved search gives its memory back and starts again
    IN PLACE (its slot keeps holding the window), the lowest running seed is never the one to yield -- the contigs, scores and
    expansion counts are those of the run with all the room, whatever starved when (advisor r2: re-runs after the pass saw other paths)"""
    from megagta_amd import api
    g, fw, rv, kmers, states = meta20k.g, meta20k.fw, meta20k.rv, meta20k.kmers, meta20k.states
    if True:
        seen_yield, seen_resume, seen_reserve, sizes = False, False, False, []
        # (pools in KB; a search that outgrows its base arena holds three 2 MB pages at least -- nodes, heap slots, hash bucket -- so the
        # small pools serve one or two searches at a time, through the reserve, resumed passes and the one-search-at-a-time last resort:
        # 4 MB cannot hold one grown search = the loud error; 8 and 12 MB finish after two and one resumed passes; 16 MB needs none and
        # serves the lowest search from the reserve alone)
        # (the 12 MB pool under window 8 is test_giving_up_the_order_is_opt_in_and_says_so's "held" run)
        for (window, rate), pools in (((8, 0), (4096, 8192)), ((64, 4), (16384,))):
            want, st0 = api.astar_search(g, fw, rv, kmers, states, 0, 0.5, cache_mode=window, cost_rate=rate)      # prune 0: the largest searches
            assert st0["n_retries"] == 0
            try:
                for kb in pools:
                    ctx.set_search_arena(7, kb << 10)
                    try:
                        got, st = api.astar_search(g, fw, rv, kmers, states, 0, 0.5, cache_mode=window, cost_rate=rate)
                    except api.MegaGtaError as e:         # a pool that cannot hold even the lowest running search alone: a loud error
                        assert "do not fit" in str(e)
                        sizes.append((window, kb, "error"))
                        continue
                    sizes.append((window, kb, st["n_retries"], st["n_resumes"], st["reserve_used"]))
                    seen_yield |= st["n_retries"] > 0
                    seen_resume |= st["n_resumes"] > 0        # the lowest search outgrew its reserve: the batch went on behind its commit frontier
                    seen_reserve |= st["reserve_used"] > 0    # the lowest search was served by the reserve
                    assert st["n_expansions"] == st0["n_expansions"], sizes
                    for a, b, km in zip(got, want, kmers):
                        assert a.contig(km) == b.contig(km) and a.right_side == b.right_side and a.left_side == b.left_side, sizes
            finally:
                ctx.set_search_arena(0, 0)
        print(sizes)
        assert seen_resume and seen_reserve, sizes            # (in-place restarts need the lowest search to call with its reserve used up: rare since round 4, reported only)


@pytest.mark.parametrize("M,k1,prune,pen,seed", [(60, 30, 20, 0.5, 1), (90, 36, 0, 0.5, 2), (150, 45, 20, 0.0, 3), (75, 45, 5, 2.0, 4),
                                                   (200, 36, 20, 0.5, 5), (48, 30, 3, 0.25, 6)])
def test_fuzz_genes_k_and_search_options_vs_oracle(ctx, oracle, tmp_path, M, k1, prune, pen, seed):
    """other gene lengths, k = 30 / 36 / 45, pruning on / off / tight, low-coverage penalties; the seeds are the ones OUR findstart reports
    on the same reads (the real pipeline), plus seeds that are absent from the graph: GPU == oracle per seed, scores bit-equal"""
    from megagta_amd import api, findstart
    mg = synth.make_metagenome(4000, 150, (("g", M),), seed=40 + seed, reads_per_genome=500, genome_len=6000)
    packed, start = synth.pack_reads_for_build(mg.reads)
    k = k1 - 1
    stream = ctx.build_sdbg(ctx.upload_reads(packed, start), k)
    synth.write_gene_models(mg.genes, str(tmp_path))
    fpath, rpath, faa = (os.path.join(str(tmp_path), "g", n) for n in ("for_enone.hmm", "rev_enone.hmm", "ref_aligned.faa"))
    lines, _ = findstart.find_start(ctx, faa, list(mg.reads), k1)
    seeds = [(l.split("\t")[3], int(l.split("\t")[7])) for l in lines][:150]
    seeds += synth.synthetic_seeds(mg.genes[0], k1, 30, seed=seed)
    assert len(seeds) > 40
    g = api.Graph(ctx, stream)
    fw, rv = api.DeviceHmm(ctx, hmmlib.parse_hmm(fpath)), api.DeviceHmm(ctx, hmmlib.parse_hmm(rpath))
    res, st = api.astar_search(g, fw, rv, [s[0] for s in seeds], [s[1] - 1 for s in seeds], prune, pen)
    og = oracle.Graph(oracle.Stream.build(packed, start, k, threads=4))
    S = oracle.Searcher(og, oracle.Hmm(fpath), oracle.Hmm(rpath), prune, pen)
    for (kmer, pos), r in zip(seeds, res):
        contig, R, L = S.search(kmer, pos - 1, cold=True)
        assert r.contig(kmer) == contig
        for got, ref in ((r.right_side, R), (r.left_side, L)):
            assert (got["ok"], got["n_closed"], got["n_expanded"], got["partial"], got["n_opened"]) == \
                   (ref.ok, ref.n_closed, ref.n_expanded, ref.partial, ref.n_opened)
            if ref.ok:
                assert got["real_score"] == ref.real_score and got["score"] == ref.score and got["fval"] == ref.fval
