"""Parity of the HIP SdBG build (through the C ABI) against the CPU oracle and the golden vectors.
Bar: the logical edge stream is BIT-EXACT (records, large multiplicities, tip labels, bucket sizes)."""
import json
import os

import numpy as np
import pytest

from megagta_amd import readlib, synth
from tests import helpers as H

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    from megagta_amd import api
    c = api.Context(0)
    yield c
    c.close()


def _same(gpu, orc):
    assert gpu.k == orc.k and gpu.words_per_tip == orc.words_per_tip
    assert np.array_equal(gpu.bucket_items, orc.bucket_items)
    assert np.array_equal(gpu.records, orc.records)
    assert np.array_equal(gpu.large, orc.large)
    assert np.array_equal(gpu.tips, orc.tips)
    assert gpu.md5() == orc.md5()


@pytest.mark.parametrize("k", [29, 35, 44])
def test_toy_vs_oracle_and_golden(ctx, oracle, golden_dir, k):
    packed, start = readlib.load_for_build(os.path.join(golden_dir, "toy", "reads.lib"))
    rd = ctx.upload_reads(packed, start)
    g = ctx.build_sdbg(rd, k)
    o = oracle.Stream.build(packed, start, k, threads=4).edges()
    _same(g, o)
    fx = H.load_streams(os.path.join(golden_dir, "toy", "sdbg_streams.json"))[str(k)]
    assert g.md5() == fx["md5"]                           # == the reference's buildgraph output
    assert g.stats["n_items"] == o.n_items_sorted
    assert g.stats["n_kmers"] == 6000 * (150 - k)
    # bucket-level tip / large counts agree with a decode of the stream
    assert g.bucket_tips.sum() == ((g.records >> 5) & 1).sum() and g.bucket_large.sum() == g.large.size


@pytest.mark.parametrize("k", [21, 29, 31, 44, 47, 63])
def test_ragged_edge_cases(ctx, golden_dir, k):
    """ragged lengths, reads shorter than k+1, N->G, palindromic (k+1)-mers, multiplicity > 254, W = 2..5 words"""
    packed, start = readlib.load_for_build(os.path.join(golden_dir, "ragged", "reads.lib"))
    g = ctx.build_sdbg(ctx.upload_reads(packed, start), k)
    fx = H.load_streams(os.path.join(golden_dir, "ragged", "sdbg_streams.json"))[str(k)]
    assert g.md5() == fx["md5"]
    assert [int(x) for x in g.records[:256]] == fx["head_records"]


@pytest.mark.parametrize("k", [15, 30, 46, 79, 95, 127])
def test_key_width_sweep(ctx, oracle, k):
    """every key width W = 2..9 incl. the widths where 2k+4 fills the last word exactly (k = 30, 46)"""
    rng = np.random.default_rng(k)
    genome = rng.integers(0, 4, 4000).astype(np.uint8)
    reads = []
    for _ in range(300):
        L = int(rng.integers(k - 3, 260))
        p = int(rng.integers(0, 4000 - L))
        r = genome[p:p + L].copy()
        if rng.random() < 0.5:
            r = (3 - r[::-1]).astype(np.uint8)
        reads.append(r)
    packed, start = readlib.pack_for_build(reads)
    g = ctx.build_sdbg(ctx.upload_reads(packed, start), k)
    o = oracle.Stream.build(packed, start, k, threads=4).edges()
    _same(g, o)


def test_empty_and_degenerate(ctx, oracle):
    # no read long enough -> empty stream, 65536 empty buckets
    reads = [np.zeros(10, np.uint8), np.ones(30, np.uint8)]
    packed, start = readlib.pack_for_build(reads)
    g = ctx.build_sdbg(ctx.upload_reads(packed, start), 44)
    assert g.records.size == 0 and g.bucket_items.sum() == 0 and g.stats["n_items"] == 0
    # a single read of exactly k+1 bases; a homopolymer (palindromic for A/T only when rc == self: poly-A vs poly-T differ)
    reads = [np.array([0, 1, 2, 3] * 8, np.uint8)[:30], np.zeros(64, np.uint8)]
    packed, start = readlib.pack_for_build(reads)
    _same(ctx.build_sdbg(ctx.upload_reads(packed, start), 29), oracle.Stream.build(packed, start, 29).edges())


def test_multi_pass_bucket_ranges(ctx, oracle, golden_dir):
    """a small memory limit forces several bucket-range passes (the analogue of CX1's lv1 loop, cx1.h:494)"""
    from megagta_amd import api
    packed, start = readlib.load_for_build(os.path.join(golden_dir, "toy", "reads.lib"))
    c2 = api.Context(0)
    c2.set_mem_limit(24 << 20)
    g = c2.build_sdbg(c2.upload_reads(packed, start), 44)
    assert g.stats["n_passes"] > 1
    fx = H.load_streams(os.path.join(golden_dir, "toy", "sdbg_streams.json"))["44"]
    assert g.md5() == fx["md5"]
    c2.close()


@pytest.mark.parametrize("m", [1, 2])
def test_count_scan_of_few_and_of_many_ranges(m):
    """one scan counts the items of every bucket range still ahead: up to 8 ranges per lane in registers, more through LDS atomics.  Both
    routes (and the key writer's side digits behind them, and -m 2's solid-run scan) give the stream of a build in one pass"""
    from megagta_amd import api
    mg = synth.make_metagenome(120_000, 150, (("rplB", 60),), seed=77)
    packed, start = synth.pack_reads_for_build(mg.reads)
    c2 = api.Context(0)
    try:
        rd = c2.upload_reads(packed, start)
        whole = c2.build_sdbg(rd, 44, min_count=m)
        assert whole.stats["n_passes"] == 1
        seen = set()
        for limit_mb in (420, 130, 48):
            c2.set_mem_limit(limit_mb << 20)
            g = c2.build_sdbg(rd, 44, min_count=m)
            _same(g, whole)
            seen.add("few" if 2 <= g.stats["n_passes"] <= 8 else ("many" if g.stats["n_passes"] > 8 else "one"))
        assert {"few", "many"} <= seen, seen
    finally:
        c2.close()


def test_larger_synthetic_properties(ctx, oracle):
    """100k x 150 bp (BASELINE config 1 size): oracle parity + size-independent properties"""
    mg = synth.make_metagenome(100_000, 150, (("rplB", 277),), seed=3)
    packed, start = synth.pack_reads_for_build(mg.reads)
    k = 29
    g = ctx.build_sdbg(ctx.upload_reads(packed, start), k)
    o = oracle.Stream.build(packed, start, k, threads=8).edges()
    _same(g, o)
    # multiplicities of all records account for every sorted item that was not $-suppressed: upper bound
    mult = (g.records >> 8).astype(np.int64)
    assert mult.sum() <= g.stats["n_items"]
    # `last` flags: one per distinct node with an outgoing a != $  => count(last) <= edges, > 0
    assert 0 < ((g.records >> 4) & 1).sum() <= g.records.size
    # W never exceeds 8; tips have last == 0
    assert (g.records & 15).max() <= 8
    tip = ((g.records >> 5) & 1).astype(bool)
    assert (((g.records >> 4) & 1)[tip] == 0).all()


def test_unsupported_is_loud():
    """what is not supported fails loudly, never falls back: k out of range"""
    from megagta_amd import api
    c = api.Context(0)
    rd = c.upload_reads(np.zeros(4, np.uint32), np.array([0, 60], np.uint64))
    with pytest.raises(api.MegaGtaError):
        c.build_sdbg(rd, 200)
    c.close()


def test_device_export_matches_host_collect(ctx, golden_dir):
    """the device-resident record shard handed to the RCCL all-gather == what the host sink receives; bucket-range shards concatenate to the whole"""
    import torch
    from megagta_amd import api
    packed, start = readlib.load_for_build(os.path.join(golden_dir, "toy", "reads.lib"))
    rd = ctx.upload_reads(packed, start)
    whole = ctx.build_sdbg(rd, 44)
    parts = []
    for r in range(2):
        b0, b1 = r * 32768, (r + 1) * 32768
        g = ctx.build_sdbg(rd, 44, bucket_range=(b0, b1))
        t = api.export_records_to_torch(ctx)
        assert np.array_equal(t.cpu().numpy().view(np.uint16), g.records)
        assert g.bucket_items[:b0].sum() == 0 and g.bucket_items[b1:].sum() == 0
        parts.append(g.records)
    assert np.array_equal(np.concatenate(parts), whole.records)


def _solid_cases(golden_dir, sub):
    d = json.load(open(os.path.join(golden_dir, sub, "sdbg_streams_solid.json")))
    out = []
    for tag, fx in d.items():
        k, m, mercy = int(tag.split("_")[0][1:]), int(tag.split("_")[1][1:]), tag.endswith("_mercy")
        out.append((tag, k, m, mercy, fx))
    return out


@pytest.mark.parametrize("sub", ["toy", "ragged"])
def test_min_count_and_mercy_vs_reference(ctx, golden_dir, sub):
    """-m >= 2: stage-1 solid-edge counting (+ mercy edges) then stage 2 == the reference's buildgraph -m M [--need_mercy]:
    edge stream bit-exact and the .counting histogram identical"""
    import hashlib
    from megagta_amd import api
    packed, start = readlib.load_for_build(os.path.join(golden_dir, sub, "reads.lib"))
    rd = ctx.upload_reads(packed, start)
    ran = 0
    for tag, k, m, mercy, fx in _solid_cases(golden_dir, sub):
        if fx.get("reference_crashed"):
            continue                                         # the reference itself segfaults on this input (recorded in the fixture)
        g = ctx.build_sdbg(rd, k, min_count=m, need_mercy=mercy)
        assert int(g.records.size) == fx["num_edges"], tag
        assert [int(x) for x in g.records[:256]] == fx["head_records"], tag
        assert g.md5() == fx["md5"], tag
        assert hashlib.md5(api.counting_text(ctx.last_counting()).encode()).hexdigest() == fx["counting_md5"], tag
        ran += 1
    assert ran >= 4


@pytest.mark.parametrize("mode", [1, 2, 3])
@pytest.mark.parametrize("k", [31, 44, 79])
def test_sort_routes_agree(ctx, oracle, golden_dir, k, mode):
    """the three ways a key can get sorted (global LSD only / LDS LSD passes / LDS passes + finish by comparison) give one stream"""
    packed, start = readlib.load_for_build(os.path.join(golden_dir, "toy", "reads.lib"))
    rd = ctx.upload_reads(packed, start)
    base = ctx.build_sdbg(rd, k)
    ctx.set_full_lsd(mode)
    try:
        g = ctx.build_sdbg(rd, k)
    finally:
        ctx.set_full_lsd(0)
    _same(g, base)
    o = oracle.Stream.build(packed, start, k, threads=4).edges()
    _same(base, o)


@pytest.mark.parametrize("k", [21, 44])
def test_redundant_reads_long_runs(ctx, oracle, k):
    """highly redundant input: runs of equal keys far longer than the comparison route accepts (per-tile fallback), next to
    unique reads (comparison route), next to a k-mer so frequent that its segment leaves LDS (deferred / global fallback)"""
    rng = np.random.default_rng(7 + k)
    reads = []
    for i in range(40):
        r = rng.integers(0, 4, 120).astype(np.uint8)
        reads += [r.copy() for _ in range(int(rng.integers(2, 400)))]
    for i in range(3000):
        r = rng.integers(0, 4, int(rng.integers(k + 1, 140))).astype(np.uint8)
        reads.append(r)
    reads += [np.zeros(150, np.uint8) for _ in range(300)]            # poly-A: one (k+1)-mer, 300 * (150 - k) copies
    order = rng.permutation(len(reads))
    reads = [reads[i] for i in order]
    packed, start = readlib.pack_for_build(reads)
    g = ctx.build_sdbg(ctx.upload_reads(packed, start), k)
    o = oracle.Stream.build(packed, start, k, threads=4).edges()
    _same(g, o)


@pytest.mark.parametrize("k,m", [(21, 1), (44, 1), (79, 1), (31, 2)])
def test_segments_between_one_and_two_lds_tiles(ctx, oracle, k, m):
    """a 16-mer prefix with 5-8 thousand keys (more than the 4096-key LDS tile, no more than twice that: sorted by a workgroup that has the
    CU's LDS to itself, round 3) next to shorter and longer ones (poly-C: one tile; poly-G: the global route), with and without stage 1"""
    rng = np.random.default_rng(11 * k + m)
    reads = []

    def homopolymer(c, copies, mixed):
        out = [np.full(150, c, np.uint8) for _ in range(copies)]
        for _ in range(mixed):                                         # the same leading characters, then something else: unequal keys of the segment
            r = rng.integers(0, 4, 150).astype(np.uint8)
            r[:k + 12] = c
            out.append(r)
        return out
    per = 150 - k
    reads += homopolymer(0, 6400 // per, 30)                           # A (and its T on the other strand): ~6.4 k equal keys + ~400 others
    reads += homopolymer(1, 2600 // per, 10)                           # C / G: fits one tile
    for _ in range(2500):
        reads.append(rng.integers(0, 4, int(rng.integers(k + 1, 160))).astype(np.uint8))
    order = rng.permutation(len(reads))
    reads = [reads[i] for i in order]
    packed, start = readlib.pack_for_build(reads)
    rd = ctx.upload_reads(packed, start)
    if m == 1:
        g = ctx.build_sdbg(rd, k)
        o = oracle.Stream.build(packed, start, k, threads=4).edges()
    else:
        g = ctx.build_sdbg(rd, k, min_count=m, need_mercy=True)
        o = oracle.Stream.build_solid(packed, start, k, m, True, threads=4).edges()
    assert g.stats["n_big_segments"] >= 2                              # the A and the T segment at least were deferred
    _same(g, o)


@pytest.mark.parametrize("k,m,mercy,assist", [(21, 2, True, 0), (31, 3, False, 12), (44, 2, True, 12), (63, 2, True, 0),
                                               (111, 2, True, 0), (120, 3, False, 6), (127, 2, True, 6)])   # k > 110: 10- / 11-word sort records (round 2)
def test_min_count_vs_oracle_seeded(ctx, oracle, k, m, mercy, assist):
    """stage 1 + stage 2 against the oracle's restatement on seeded reads no golden covers: coverage ~8x with substitution errors
    (solid and non-solid stretches, mercy gaps), ragged lengths, optional assist sequences (always solid, reads >= n_short)"""
    rng = np.random.default_rng(100 * k + m)
    genome = rng.integers(0, 4, 6000).astype(np.uint8)
    reads = []
    for _ in range(700):
        L = int(rng.integers(k - 2, 200))
        p = int(rng.integers(0, genome.size - L))
        r = genome[p:p + L].copy()
        err = rng.random(L) < 0.01
        r[err] = (r[err] + rng.integers(1, 4, int(err.sum()))) & 3
        if rng.random() < 0.5:
            r = (3 - r[::-1]).astype(np.uint8)
        reads.append(r)
    for _ in range(assist):
        p = int(rng.integers(0, genome.size - 400))
        reads.append(genome[p:p + 400].copy())
    packed, start = readlib.pack_for_build(reads)
    n_short = len(reads) - assist
    g = ctx.build_sdbg(ctx.upload_reads(packed, start), k, min_count=m, need_mercy=mercy, n_short_reads=n_short)
    o = oracle.Stream.build_solid(packed, start, k, m, mercy, n_short=n_short, threads=4)
    _same(g, o.edges())
    assert np.array_equal(ctx.last_counting(), o.counting)
    assert 0 < g.records.size < oracle.Stream.build(packed, start, k, threads=4).edges().records.size    # the filter removed something


@pytest.mark.parametrize("k,m,max_len", [(31, 2, 3000), (63, 2, 1500), (44, 3, 9000)])
def test_mercy_with_reads_longer_than_1024_bases(ctx, oracle, k, m, max_len):
    """--need_mercy on reads of several thousand bases (round 1 refused them: the per-read flags sat in LDS; they now go to device
    memory above 1024 bases): stage 1 + mercy + stage 2 == the oracle's restatement, mixed with short reads"""
    rng = np.random.default_rng(7 * k + max_len)
    genome = rng.integers(0, 4, 20000).astype(np.uint8)
    reads = []
    for i in range(260):
        L = int(rng.integers(k - 2, 220)) if i % 3 else int(rng.integers(1025, max_len))
        p = int(rng.integers(0, genome.size - L))
        r = genome[p:p + L].copy()
        err = rng.random(L) < 0.006
        r[err] = (r[err] + rng.integers(1, 4, int(err.sum()))) & 3
        if rng.random() < 0.5:
            r = (3 - r[::-1]).astype(np.uint8)
        reads.append(r)
    assert max(r.size for r in reads) > 1024
    packed, start = readlib.pack_for_build(reads)
    g = ctx.build_sdbg(ctx.upload_reads(packed, start), k, min_count=m, need_mercy=True)
    o = oracle.Stream.build_solid(packed, start, k, m, True, threads=4)
    _same(g, o.edges())
    assert np.array_equal(ctx.last_counting(), o.counting)
    no_mercy = oracle.Stream.build_solid(packed, start, k, m, False, threads=4).edges()
    assert g.records.size > no_mercy.records.size                     # mercy edges really were added


def test_fourth_leading_byte_and_wide_run_prefix(ctx, oracle):
    """a bucket range so narrow and so full that three leading key bytes leave segments of > 700 keys: the sort takes a fourth global
    pass and the run prefix of the LDS tiles reaches into the second key word (the regime of memory-bound passes over 10^10 items)"""
    rng = np.random.default_rng(5)
    X = np.array([0, 1, 2, 3, 0, 1, 2, 3], dtype=np.uint8)                  # ACGTACGT: every 16th position starts a key of bucket bx
    reads = []
    for _ in range(22000):
        r = rng.integers(0, 4, 160).astype(np.uint8)
        off = int(rng.integers(0, 16))
        for p in range(off, 160 - 8, 16):
            r[p:p + 8] = X
        reads.append(r[:int(rng.integers(140, 161))])
    packed, start = readlib.pack_for_build(reads)
    k = 44
    o = oracle.Stream.build(packed, start, k, threads=4).edges()
    bx = int(np.argmax(o.bucket_items))                                      # the bucket of the planted word (as the build sees the reads)
    g = ctx.build_sdbg(ctx.upload_reads(packed, start), k, bucket_range=(bx, bx + 1))
    # four leading bytes — or, since the global passes of a bucket sub-range skip the leading bits every key of the range shares
    # (MGTA_SORT_BIAS != 0, the default), two digits right below the 16 bucket bits: a 32-bit prefix either way
    assert g.stats["n_sort_launches"] == (4 if os.environ.get("MGTA_SORT_BIAS") == "0" else 2) and g.stats["n_items"] > 180_000
    lo, hi = int(o.bucket_items[:bx].sum()), int(o.bucket_items[:bx + 1].sum())
    assert hi - lo == g.records.size > 10_000
    assert np.array_equal(g.records, o.records[lo:hi])
    assert g.bucket_items[bx] == hi - lo and g.bucket_items.sum() == hi - lo


@pytest.mark.parametrize("k,L", [(29, 100), (60, 150), (95, 250), (127, 250)])
def test_routes_agree_at_scale_other_key_widths(ctx, k, L):
    """10^7..10^8 sort items, W = 2, 4, 6, 8 key words, read lengths 100..250: the comparison finish and the LSD finish of the LDS
    tiles (independent code after the tile prologue) produce the same stream, and so does a build split over three bucket ranges"""
    from megagta_amd import synth
    mg = synth.make_metagenome(200_000, L, (("rplB", 60),), seed=k)
    packed, start = synth.pack_reads_for_build(mg.reads)
    rd = ctx.upload_reads(packed, start)
    a = ctx.build_sdbg(rd, k)
    assert a.stats["n_lsd_tiles"] * 4 < a.stats["n_items"] / 4096          # the comparison route did the work
    ctx.set_full_lsd(2)
    try:
        b = ctx.build_sdbg(rd, k)
    finally:
        ctx.set_full_lsd(0)
    _same(a, b)
    parts = [ctx.build_sdbg(rd, k, bucket_range=r) for r in ((0, 20000), (20000, 47000), (47000, 65536))]
    assert np.array_equal(np.concatenate([p.records for p in parts]), a.records)
    assert np.array_equal(np.concatenate([p.tips for p in parts]), a.tips)
    assert np.array_equal(sum(p.bucket_items for p in parts), a.bucket_items)


@pytest.mark.parametrize("k,L,m", [(29, 100, 1), (44, 150, 1), (95, 250, 1), (127, 250, 1), (44, 150, 2)])
def test_side_digit_census_equals_the_census_of_the_keys(ctx, monkeypatch, k, L, m):
    """the scatter of a global pass leaves the next pass's digit of every key in a byte array and that pass counts the bytes instead of
    reading the keys: with MGTA_SORT_SIDE=2 every such census is repeated from the keys and compared (the build fails if they differ), over
    W = 2, 3, 6, 8 key words, the payload-carrying keys of -m 2, whole-range and sub-range (biased digits) passes; and the stream is the
    one of a build without side digits"""
    from megagta_amd import synth
    mg = synth.make_metagenome(150_000, L, (("rplB", 60),), seed=300 + k)
    packed, start = synth.pack_reads_for_build(mg.reads)
    rd = ctx.upload_reads(packed, start)
    monkeypatch.setenv("MGTA_SORT_SIDE", "0")
    plain = ctx.build_sdbg(rd, k, min_count=m)
    assert plain.stats["n_sort_launches"] >= 2
    monkeypatch.setenv("MGTA_SORT_SIDE", "2")
    _same(ctx.build_sdbg(rd, k, min_count=m), plain)
    if m == 1:
        monkeypatch.setenv("MGTA_SORT_BIAS", "2")
        parts = [ctx.build_sdbg(rd, k, bucket_range=r) for r in ((0, 30000), (30000, 65536))]
        assert np.array_equal(np.concatenate([p.records for p in parts]), plain.records)
        monkeypatch.delenv("MGTA_SORT_BIAS")
    monkeypatch.delenv("MGTA_SORT_SIDE")
    _same(ctx.build_sdbg(rd, k, min_count=m), plain)


@pytest.mark.parametrize("k,m", [(29, 2), (60, 3)])
def test_stage1_routes_agree_at_scale(ctx, k, m):
    """-m >= 2 on 3*10^7 stage-1 sort items (payload-carrying keys, masked comparisons): comparison finish == LSD finish, stream and
    .counting histogram alike"""
    from megagta_amd import synth
    mg = synth.make_metagenome(200_000, 150, (("rplB", 60),), seed=100 + k)
    packed, start = synth.pack_reads_for_build(mg.reads)
    rd = ctx.upload_reads(packed, start)
    a = ctx.build_sdbg(rd, k, min_count=m, need_mercy=True)
    ca = ctx.last_counting().copy()
    ctx.set_full_lsd(2)
    try:
        b = ctx.build_sdbg(rd, k, min_count=m, need_mercy=True)
        cb = ctx.last_counting().copy()
    finally:
        ctx.set_full_lsd(0)
    _same(a, b)
    assert np.array_equal(ca, cb) and ca.sum() > 0
    assert 0 < a.records.size < ctx.build_sdbg(rd, k).records.size


@pytest.mark.parametrize("k,length,n_reads", [(20, 21, 333), (20, 22, 129), (31, 31 + 4096, 70), (31, 31 + 4097, 70), (44, 150, 1000), (44, 150, 63), (27, 91, 257)])
def test_reads_of_one_length_sub_range_scan_vs_oracle(ctx, oracle, k, length, n_reads):
    """the scans of a bucket sub-range number the positions of a wave's reads through when they are all of one length (idx / npos by a
    multiplication): one position per read, two, 4096 (the limit of that route) and 4097 (a read at a time), a last workgroup with fewer
    than 64 reads, a wave with one read; every range against the oracle's stream"""
    rng = np.random.default_rng(5000 + k + length)
    genome = rng.integers(0, 4, length * 3 + 50).astype(np.uint8)
    reads = []
    for i in range(n_reads):
        p = int(rng.integers(0, genome.size - length + 1))
        r = genome[p:p + length].copy()
        if i % 3 == 0:
            r = (3 - r[::-1]).astype(np.uint8)
        if i % 7 == 0:
            r[int(rng.integers(0, length))] = int(rng.integers(0, 4))
        reads.append(r)
    packed, start = readlib.pack_for_build(reads)
    rd = ctx.upload_reads(packed, start)
    o = oracle.Stream.build(packed, start, k, threads=4).edges()
    for b0, b1 in ((0, 65536), (0, 20000), (20000, 65536), (12345, 12346 + 30000)):
        g = ctx.build_sdbg(rd, k, bucket_range=(b0, b1))
        lo, hi = int(o.bucket_items[:b0].sum()), int(o.bucket_items[:b1].sum())
        assert np.array_equal(g.records, o.records[lo:hi]), (b0, b1)
        assert np.array_equal(g.bucket_items[b0:b1], o.bucket_items[b0:b1]), (b0, b1)
    ctx.set_full_lsd(1)                                                      # (no closed form: the whole range through the scans too)
    try:
        _same(ctx.build_sdbg(rd, k), o)
    finally:
        ctx.set_full_lsd(0)


@pytest.mark.parametrize("seed", list(range(64)))
def test_fuzz_small_inputs_vs_oracle(ctx, oracle, seed):
    """seeded random inputs: any k in [9, 127], ragged read lengths (also shorter than k+1), duplicated and reverse-complemented reads,
    low-complexity stretches (palindromes, hot k-mers), a random bucket sub-range, and -m 2/3 with mercy on every third seed"""
    rng = np.random.default_rng(1000 + seed)
    k = int(rng.choice([9, 10, 15, 16, 20, 27, 31, 32, 44, 47, 48, 63, 64, 80, 95, 111, 127]))
    genome = rng.integers(0, 4, int(rng.integers(300, 3000))).astype(np.uint8)
    if seed % 4 == 0:
        genome[100:160] = 0                                                   # poly-A
        genome[200:260] = np.tile(np.array([0, 3], dtype=np.uint8), 30)        # (AT)n: palindromic (k+1)-mers when k+1 is even
    reads = []
    for _ in range(int(rng.integers(20, 400))):
        L = int(rng.integers(1, min(genome.size, 330)))
        p = int(rng.integers(0, genome.size - L + 1))
        r = genome[p:p + L].copy()
        if rng.random() < 0.1 and L > 3:
            r[int(rng.integers(0, L))] = int(rng.integers(0, 4))
        if rng.random() < 0.5:
            r = (3 - r[::-1]).astype(np.uint8)
        reads.append(r)
        if rng.random() < 0.2:
            reads.append(r.copy())
    packed, start = readlib.pack_for_build(reads)
    rd = ctx.upload_reads(packed, start)
    g = ctx.build_sdbg(rd, k)
    o = oracle.Stream.build(packed, start, k, threads=2).edges()
    _same(g, o)
    b0 = int(rng.integers(0, 65535))
    b1 = int(rng.integers(b0 + 1, 65537))
    part = ctx.build_sdbg(rd, k, bucket_range=(b0, b1))
    lo, hi = int(o.bucket_items[:b0].sum()), int(o.bucket_items[:b1].sum())
    assert np.array_equal(part.records, o.records[lo:hi]) and np.array_equal(part.bucket_items[b0:b1], o.bucket_items[b0:b1])
    if seed % 3 == 0:
        m = int(rng.choice([2, 3]))
        mercy = True
        gs = ctx.build_sdbg(rd, k, min_count=m, need_mercy=mercy)
        os_ = oracle.Stream.build_solid(packed, start, k, m, mercy, threads=2)
        _same(gs, os_.edges())
        assert np.array_equal(ctx.last_counting(), os_.counting)


@pytest.mark.parametrize("k", [30, 44])
def test_tiled_key_writer_many_tiles_vs_oracle(ctx, oracle, k):
    """closed-form builds (k+1 odd, -m 1, every bucket) write the keys tile by tile of the first sort pass and count that pass's
    census on the way: tens of tiles, reads cut by tile bounds, reads longer than a tile, whole groups of 64 reads too short to
    yield a key, a ragged tail"""
    rng = np.random.default_rng(1000 + k)
    genome = rng.integers(0, 4, 200_000).astype(np.uint8)
    reads = []

    def take(L):
        p = int(rng.integers(0, genome.size - L))
        r = genome[p:p + L].copy()
        return (3 - r[::-1]).astype(np.uint8) if rng.random() < 0.5 else r

    for i in range(9000):
        if 2000 <= i < 2200 or 5000 <= i < 5064:
            reads.append(take(int(rng.integers(1, k + 1))))            # no key at all (>= one whole group of 64 reads)
        else:
            reads.append(take(int(rng.integers(k - 2, 260))))
    reads.insert(3000, take(40_000))                                    # longer than a tile of 32768 keys
    reads.insert(3001, take(17_000))
    packed, start = readlib.pack_for_build(reads)
    g = ctx.build_sdbg(ctx.upload_reads(packed, start), k)
    assert g.stats["n_items"] > 40 * 32768
    o = oracle.Stream.build(packed, start, k, threads=8).edges()
    _same(g, o)


@pytest.mark.parametrize("k", [21, 29, 35, 43])
def test_closed_form_even_k1_palindromes_vs_oracle(ctx, oracle, k):
    """k+1 even: a (k+1)-mer can equal its reverse complement (s2.cpp:278: forward items only).  The closed-form key layout keeps the rc
    slot of such a position and fills it with a sentinel key that sorts behind every real key; (AT)n / (CG)n / (ACGT)n stretches, whole
    reads of them, stretches at read ends (the $ items) and poly-T next to the sentinels' prefix"""
    rng = np.random.default_rng(7000 + k)
    reads = []
    at, cg, acgt = np.array([0, 3], np.uint8), np.array([1, 2], np.uint8), np.array([0, 1, 2, 3], np.uint8)
    for i in range(6000):
        L = int(rng.integers(k - 2, 220))
        r = rng.integers(0, 4, L).astype(np.uint8)
        u = rng.random()
        if u < 0.25:
            unit = (at, cg, acgt)[int(rng.integers(0, 3))]
            n = int(rng.integers(k + 1, max(k + 2, L)))
            p = int(rng.integers(0, max(1, L - n + 1))) if rng.random() < 0.6 else (0 if rng.random() < 0.5 else max(0, L - n))
            rep = np.tile(unit, n // unit.size + 2)[int(rng.integers(0, unit.size)):][:min(n, L - p)]
            r[p:p + rep.size] = rep
        elif u < 0.3:
            r[:] = 3                                                    # poly-T: shares every leading byte with the sentinel
        elif u < 0.35:
            r = np.tile(at, L)[:L].copy()                               # every position palindromic
        reads.append(r)
    packed, start = readlib.pack_for_build(reads)
    g = ctx.build_sdbg(ctx.upload_reads(packed, start), k)
    o = oracle.Stream.build(packed, start, k, threads=8).edges()
    _same(g, o)
