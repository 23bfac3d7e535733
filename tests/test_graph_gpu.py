"""Device graph (load + rank/select navigation) vs the oracle and the reference's own answers."""
import os

import numpy as np
import pytest

from megagta_amd import readlib
from tests import helpers as H

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    from megagta_amd import api
    c = api.Context(0)
    yield c
    c.close()


def _graphs(ctx, oracle, lib_prefix, k):
    from megagta_amd import api
    packed, start = readlib.load_for_build(lib_prefix)
    stream = ctx.build_sdbg(ctx.upload_reads(packed, start), k)
    og = oracle.Graph(oracle.Stream.build(packed, start, k, threads=4))
    return api.Graph(ctx, stream), og


def test_outgoing_all_edges_toy(ctx, oracle, golden_dir):
    g, og = _graphs(ctx, oracle, os.path.join(golden_dir, "toy", "reads.lib"), 44)
    assert g.size == og.size
    rng = np.random.default_rng(1)
    ids = np.concatenate([np.arange(0, 3000), rng.integers(0, g.size, 20000), np.arange(g.size - 3000, g.size)])
    deg, out = g.outgoing(ids)
    for e, d, o in zip(ids.tolist(), deg.tolist(), out.tolist()):
        n, ref = og.outgoing(e)
        assert d == n and o[:max(n, 0)] == ref, e
    # and the reference's own answers (probe)
    _, qs = H.parse_probe_graph(H.gz_lines(os.path.join(golden_dir, "toy", "graph_k44.txt.gz")))
    deg, out = g.outgoing([q["e"] for q in qs])
    for q, d, o in zip(qs, deg.tolist(), out.tolist()):
        assert d == q["od"] and o[:max(d, 0)] == q["out"]


@pytest.mark.parametrize("k", [29, 47])
def test_outgoing_ragged_every_edge(ctx, oracle, golden_dir, k):
    """small graphs with tips, $ edges, large multiplicities: every edge id"""
    g, og = _graphs(ctx, oracle, os.path.join(golden_dir, "ragged", "reads.lib"), k)
    ids = np.arange(g.size)
    deg, out = g.outgoing(ids)
    for e in range(g.size):
        n, ref = og.outgoing(e)
        assert deg[e] == n and out[e, :max(n, 0)].tolist() == ref, e
    _, qs = H.parse_probe_graph(H.gz_lines(os.path.join(golden_dir, "ragged", f"graph_k{k}.txt.gz")))
    deg, out = g.outgoing([q["e"] for q in qs])
    for q, d, o in zip(qs, deg.tolist(), out.tolist()):
        assert d == q["od"] and o[:max(d, 0)] == q["out"]


def test_index_edges(ctx, oracle, golden_dir):
    g, og = _graphs(ctx, oracle, os.path.join(golden_dir, "toy", "reads.lib"), 44)
    lines = [l.split() for l in H.gz_lines(os.path.join(golden_dir, "toy", "index_k44.txt.gz"))]
    ids = g.index_edges([l[0] for l in lines])
    assert ids.tolist() == [int(l[1]) for l in lines]
    # labels of random existing edges map back to themselves or to the edge of the same node with that out-label
    rng = np.random.default_rng(2)
    kmers, want = [], []
    for e in rng.integers(0, g.size, 400).tolist():
        n, outs = og.outgoing(e)
        if n <= 0:
            continue
        lab = og.label(e)
        if "$" in lab:
            continue
        for o in outs:
            w = (og.bitvectors()["w"][o >> 4] >> np.uint64((o & 15) * 4)) & np.uint64(15) if False else None
        kmers.append(lab)
    # k-mers (node labels) + each possible next char: compare with the oracle's own search
    q = [lab + c for lab in kmers[:100] for c in "ACGT"]
    ids = g.index_edges(q)
    assert ids.tolist() == [og.index_edge(s) for s in q]


def test_empty_graph(ctx):
    from megagta_amd import api
    s = api.EdgeStream(k=29, words_per_tip=2, bucket_items=np.zeros(65536, np.int64), records=np.zeros(0, np.uint16),
                       large=np.zeros(0, np.uint16), tips=np.zeros(0, np.uint32))
    g = api.Graph(ctx, s)
    assert g.size == 0
    assert g.index_edges(["A" * 30]).tolist() == [-1]


@pytest.mark.parametrize("case,k", [("toy", 44), ("ragged", 29)])
def test_graph_from_device_resident_stream(ctx, golden_dir, case, k):
    """row f-4: the graph built from the edge stream where the build left it (no host round trip) answers like the one loaded from host
    records, on every edge; and it is refused when the last build did not leave a whole stream"""
    from megagta_amd import api
    packed, start = readlib.load_for_build(os.path.join(golden_dir, case, "reads.lib"))
    rd = ctx.upload_reads(packed, start)
    stream = ctx.build_sdbg(rd, k)
    g_host = api.Graph(ctx, stream)
    ctx.build_sdbg(rd, k, collect=False)                   # nothing handed to the host at all
    g_dev = api.Graph(ctx, None, k)
    assert g_dev.size == g_host.size == stream.records.size
    ids = np.arange(g_host.size)
    d0, o0 = g_host.outgoing(ids)
    d1, o1 = g_dev.outgoing(ids)
    assert np.array_equal(d0, d1) and np.array_equal(o0, o1)
    # index lookups go through the tip labels, which also stayed on the device
    _, qs = H.parse_probe_graph(H.gz_lines(os.path.join(golden_dir, case, f"graph_k{k}.txt.gz")))
    deg, out = g_dev.outgoing([q["e"] for q in qs])
    for q, d, o in zip(qs, deg.tolist(), out.tolist()):
        assert d == q["od"] and o[:max(d, 0)] == q["out"]
    ctx.build_sdbg(rd, k, collect=False, bucket_range=(0, 30000))
    with pytest.raises(api.MegaGtaError):
        api.Graph(ctx, None, k)


def test_graph_from_resident_stream_of_a_multi_pass_build(ctx, golden_dir):
    """row f-4 at any size: with mgta_ctx_keep_stream a build that memory forces into several bucket-range passes still leaves its whole
    edge stream on the device (each pass appended); the graph built from it answers like the host-loaded one on every edge"""
    from megagta_amd import api
    packed, start = readlib.load_for_build(os.path.join(golden_dir, "toy", "reads.lib"))
    rd = ctx.upload_reads(packed, start)
    stream = ctx.build_sdbg(rd, 44)
    g_host = api.Graph(ctx, stream)
    try:
        ctx.set_mem_limit(12 << 20)                        # the toy reads then need several bucket-range passes
        st = ctx.build_sdbg(rd, 44, collect=False).stats
        assert st["n_passes"] >= 3
        with pytest.raises(api.MegaGtaError):              # without the switch only the last pass is on the device
            api.Graph(ctx, None, 44)
        ctx.keep_stream(True)
        st = ctx.build_sdbg(rd, 44, collect=False).stats
        assert st["n_passes"] >= 3 and st["n_edges"] == stream.records.size
        g_dev = api.Graph(ctx, None, 44)
        # ... and with the lines packed INTO the stream's buffer (what a graph too large to exist twice takes: MGTA_LOAD_INPLACE forces it
        # here): the same graph, and the stream is consumed -- a second graph from it is refused
        os.environ["MGTA_LOAD_INPLACE"] = "1"
        try:
            g_inplace = api.Graph(ctx, None, 44)
        finally:
            os.environ.pop("MGTA_LOAD_INPLACE", None)
        with pytest.raises(api.MegaGtaError):
            api.Graph(ctx, None, 44)
    finally:
        ctx.set_mem_limit(0)
        ctx.keep_stream(False)
    assert g_dev.size == g_host.size == g_inplace.size
    ids = np.arange(g_host.size)
    d0, o0 = g_host.outgoing(ids)
    d1, o1 = g_dev.outgoing(ids)
    assert np.array_equal(d0, d1) and np.array_equal(o0, o1)
    d2, o2 = g_inplace.outgoing(ids)
    assert np.array_equal(d0, d2) and np.array_equal(o0, o2) and np.array_equal(g_inplace.invalid_bits(), g_host.invalid_bits())
    _, qs = H.parse_probe_graph(H.gz_lines(os.path.join(golden_dir, "toy", "graph_k44.txt.gz")))
    deg, out = g_dev.outgoing([q["e"] for q in qs])
    for q, d, o in zip(qs, deg.tolist(), out.tolist()):
        assert d == q["od"] and o[:max(d, 0)] == q["out"]


def test_shard_of_a_multi_pass_build_stays_whole_on_the_device(ctx, golden_dir):
    """the shard a rank hands to the all-gather is the edge stream of ITS bucket range -- of every pass when memory forces
    several: with mgta_ctx_keep_stream the records exported device to device equal the ones the sink collected for that range"""
    from megagta_amd import api
    packed, start = readlib.load_for_build(os.path.join(golden_dir, "toy", "reads.lib"))
    rd = ctx.upload_reads(packed, start)
    lo, hi = api.NUM_BUCKETS // 3, api.NUM_BUCKETS
    try:
        ctx.set_mem_limit(12 << 20)
        want = ctx.build_sdbg(rd, 44, bucket_range=(lo, hi))
        assert want.stats["n_passes"] >= 2
        ctx.keep_stream(True)
        st = ctx.build_sdbg(rd, 44, collect=False, bucket_range=(lo, hi)).stats
        assert st["n_passes"] >= 2
        got = api.export_records_to_torch(ctx).cpu().numpy().view(np.uint16)
        with pytest.raises(api.MegaGtaError):              # a sub-range is not a graph
            api.Graph(ctx, None, 44)
    finally:
        ctx.set_mem_limit(0)
        ctx.keep_stream(False)
    assert got.size == want.records.size and np.array_equal(got, want.records)


@pytest.mark.parametrize("seed", list(range(12)))
def test_fuzz_navigation_vs_oracle(ctx, oracle, seed):
    """random small graphs (any k, tips, $ edges, hot k-mers): OutgoingEdges of every edge and IndexBinarySearchEdge of present and absent
    (k+1)-mers agree with the oracle; so does the graph read off the device-resident stream"""
    from megagta_amd import api
    rng = np.random.default_rng(7000 + seed)
    k = int(rng.choice([9, 15, 21, 31, 32, 44, 63, 64, 95]))
    genome = rng.integers(0, 4, int(rng.integers(400, 2500))).astype(np.uint8)
    reads = []
    for _ in range(int(rng.integers(30, 250))):
        L = int(rng.integers(k + 1, min(genome.size, 300)))
        p = int(rng.integers(0, genome.size - L + 1))
        r = genome[p:p + L].copy()
        if rng.random() < 0.15:
            r[int(rng.integers(0, L))] = int(rng.integers(0, 4))
        reads.append((3 - r[::-1]).astype(np.uint8) if rng.random() < 0.5 else r)
    packed, start = readlib.pack_for_build(reads)
    stream = ctx.build_sdbg(ctx.upload_reads(packed, start), k)
    og = oracle.Graph(oracle.Stream.build(packed, start, k, threads=2))
    for g in (api.Graph(ctx, stream), api.Graph(ctx, None, k)):
        assert g.size == og.size
        deg, out = g.outgoing(np.arange(g.size))
        for e in range(g.size):
            n, ref = og.outgoing(e)
            assert deg[e] == n and out[e, :max(n, 0)].tolist() == ref, (seed, e)
        kmers = []
        for _ in range(60):
            r = reads[int(rng.integers(0, len(reads)))]
            p = int(rng.integers(0, r.size - k))
            s = r[p:p + k + 1].copy()
            if rng.random() < 0.3:
                s[int(rng.integers(0, k + 1))] = int(rng.integers(0, 4))
            kmers.append("".join("ACGT"[x] for x in s))
        ids = g.index_edges(kmers)
        want = [og.index_edge(km) for km in kmers]
        assert ids.tolist() == want


@pytest.mark.gpu
def test_forward_of_minus_edge_before_first_plain_symbol_of_its_line(ctx, oracle):
    """an edge with W = a + 4 that precedes every plain a of its 64-edge line forwards to the target of the last plain a of an EARLIER line,
    which can lie in a line before the one the line's own first plain a points to (1 edge in 74 320 here; found by the denovo parity runs).
    Every edge of the graph is checked."""
    from megagta_amd import api, synth
    reads = synth.make_strain_mix(906, n_genomes=3, genome_len=2468, read_len=100, snp_every=150)
    packed, start = readlib.pack_for_build(reads)
    st = oracle.Stream.build(packed, start, 44, threads=4)
    og, g = oracle.Graph(st), api.Graph(ctx, st.edges())
    deg, out = g.outgoing(np.arange(g.size))
    w = og.bitvectors()["w"]
    W = lambda e: (int(w[e >> 4]) >> ((e & 15) * 4)) & 15
    needed = 0
    for e in range(g.size):
        n, ref = og.outgoing(e)
        assert deg[e] == n and out[e, :max(n, 0)].tolist() == ref, e
        c = W(e)
        if c > 4 and not any(W(x) == c - 4 for x in range(e & ~63, e)):
            nxt = next((x for x in range(e, og.size) if W(x) == c - 4), None)
            needed += nxt is not None and (og.forward(e) >> 6) < (og.forward(nxt) >> 6)
    assert needed >= 1


def test_stream_kept_on_the_device_only_and_downloaded_in_one_piece(ctx, golden_dir):
    """mgta_ctx_keep_stream(ctx, 2) + mgta_sdbg_stream_detach / mgta_stream_download (what `megagta buildgraph` does in the driver's
    worker): a build forced into several bucket-range passes hands the sink counts and large multiplicities only, the graph is packed from
    the resident stream, and the stream downloaded afterwards is the one an ordinary build hands over pass by pass"""
    from megagta_amd import api
    packed, start = readlib.load_for_build(os.path.join(golden_dir, "ragged", "reads.lib"))
    rd = ctx.upload_reads(packed, start)
    want = ctx.build_sdbg(rd, 29)
    try:
        ctx.set_mem_limit(6 << 20)                                     # several passes
        ctx.keep_stream(2)
        got = ctx.build_sdbg(rd, 29)
        assert got.stats["n_passes"] > 1 and got.records.size == 0 and got.tips.size == 0      # nothing crossed the bus pass by pass
        assert np.array_equal(got.bucket_items, want.bucket_items) and np.array_equal(got.large, want.large)
        g = api.Graph(ctx, None, 29)                                   # packed from the resident stream
        assert g.size == want.records.size
        recs, tips = ctx.detach_stream()
        assert np.array_equal(recs, want.records) and np.array_equal(tips, want.tips)
        with pytest.raises(api.MegaGtaError):                          # the context has forgotten the stream
            ctx.detach_stream()
        g.free()
    finally:
        ctx.keep_stream(False)
        ctx.set_mem_limit(0)
