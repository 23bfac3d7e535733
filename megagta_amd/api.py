"""Python host side above the C ABI: mirrors the reference's `buildgraph` / `search` operator
surface (same inputs, same outputs, same error behaviour) on top of libmegagta_hip.so.

  Context.build_sdbg(...)      <->  `megagta buildgraph`  (build_graph.cpp:33-135)
  write_sdbg(prefix, stream)   <->  SdbgWriter            (sdbg_multi_io.h:34-199)
  read_sdbg(prefix)            <->  SdbgReader            (sdbg_multi_io.h:201-417)
"""
from __future__ import annotations

import ctypes as C
import hashlib
import os
from dataclasses import dataclass, field

import numpy as np

from . import _lib
from ._lib import MegaGtaError, check

NUM_BUCKETS = 65536


@dataclass
class EdgeStream:
    """Logical SdBG edge stream in bucket order (what SdbgWriter::write receives)."""
    k: int
    words_per_tip: int
    bucket_items: np.ndarray                 # int64 [65536] records per bucket
    records: np.ndarray                      # uint16 [num_edges]
    large: np.ndarray                        # uint16 [num_large]
    tips: np.ndarray                         # uint32 [num_tips * words_per_tip]
    bucket_large: np.ndarray = None          # int64 [65536]
    bucket_tips: np.ndarray = None           # int64 [65536]
    stats: dict = field(default_factory=dict)

    def md5(self) -> str:
        h = hashlib.md5()
        h.update(np.int32(self.k).tobytes())
        h.update(self.bucket_items.astype("<i8").tobytes())
        h.update(self.records.astype("<u2").tobytes())
        h.update(self.large.astype("<u2").tobytes())
        h.update(self.tips.astype("<u4").tobytes())
        return h.hexdigest()


class Context:
    """One per GPU (mgta_ctx)."""

    def __init__(self, device: int = 0):
        self._L = _lib.load()
        self.h = self._L.mgta_ctx_create(device)
        if not self.h:
            raise MegaGtaError("mgta_ctx_create failed: " + self._L.mgta_last_error().decode())

    def close(self):
        if getattr(self, "h", None):
            self._L.mgta_ctx_destroy(self.h)
            self.h = None

    __del__ = close

    def set_full_lsd(self, on: int):
        """diagnostic bit mask: 1 = global LSD passes only (no segment-local LDS finish); 2 = LDS finish by LSD passes only (no comparison route)"""
        check(self._L.mgta_ctx_set_full_lsd(self.h, int(on)), "mgta_ctx_set_full_lsd")

    def last_counting(self) -> np.ndarray:
        """(k+1)-mer multiplicity histogram of the last min_count >= 2 build (int64[65536]); `counting_text` renders PREFIX.counting"""
        h = np.zeros(65536, dtype=np.int64)
        check(self._L.mgta_sdbg_last_counting(self.h, h.ctypes.data), "mgta_sdbg_last_counting")
        return h

    def set_mem_limit(self, nbytes: int):
        check(self._L.mgta_ctx_set_mem_limit(self.h, nbytes), "mgta_ctx_set_mem_limit")

    def probe_random_lines(self, table_bytes: int, configs) -> list[dict]:
        """The device's random 128-byte line rate / dependent-line latency (mgta_probe_random_lines).  configs: iterable of
        (waves_per_cu, groups, unroll, dependent, steps); one dict per configuration (lines_in_flight_per_cu, gb_per_s, ns_per_step, ...)."""
        configs = list(configs)
        arr = (_lib.LineProbe * len(configs))()
        for c, (w, g, u, d, steps) in zip(arr, configs):
            c.waves_per_cu, c.groups, c.unroll, c.dependent, c.steps = int(w), int(g), int(u), int(d), int(steps)
        check(self._L.mgta_probe_random_lines(self.h, int(table_bytes), arr, len(configs)), "mgta_probe_random_lines")
        return [c.as_dict() for c in arr]

    def keep_stream(self, on=True):
        """whole-range builds leave their whole edge stream on the device, also when they take several memory-bound passes.  on = 2: and the
        records / tip labels of a pass are not copied to the host for the sink (`detach_stream` brings the whole stream over afterwards)"""
        check(self._L.mgta_ctx_keep_stream(self.h, int(on)), "mgta_ctx_keep_stream")

    def detach_stream(self) -> tuple[np.ndarray, np.ndarray]:
        """the whole edge stream of the last keep-stream build: taken out of the context (mgta_sdbg_stream_detach), downloaded in one piece
        (mgta_stream_download) and freed -> (records uint16, tip label words uint32)"""
        h = C.c_void_p()
        check(self._L.mgta_sdbg_stream_detach(self.h, C.byref(h)), "mgta_sdbg_stream_detach")
        try:
            nr, nt = C.c_uint64(), C.c_uint64()
            check(self._L.mgta_stream_sizes(h, C.byref(nr), C.byref(nt)), "mgta_stream_sizes")
            recs, tips = np.empty(nr.value, dtype=np.uint16), np.empty(nt.value, dtype=np.uint32)
            check(self._L.mgta_stream_download(h, recs.ctypes.data, tips.ctypes.data), "mgta_stream_download")
        finally:
            self._L.mgta_stream_free(h)
        return recs, tips

    def release_scratch(self):
        """free the work memory kept between calls (build pool, search pool)"""
        check(self._L.mgta_ctx_release_scratch(self.h), "mgta_ctx_release_scratch")

    def set_search_arena(self, log2_base_nodes: int = 0, pool_bytes: int = 0):
        """work memory of the A* searches: base arena of 1 << log2_base_nodes nodes per search slot (0 = default), pool the searches
        grow into (0 = auto).  Small values exercise the in-place growth on small inputs."""
        check(self._L.mgta_ctx_set_search_arena(self.h, int(log2_base_nodes), int(pool_bytes)), "mgta_ctx_set_search_arena")

    # ---- SdBG build ---------------------------------------------------------------------------
    def upload_reads(self, packed: np.ndarray, start_idx: np.ndarray) -> "Reads":
        packed = np.ascontiguousarray(packed, dtype=np.uint32)
        start_idx = np.ascontiguousarray(start_idx, dtype=np.uint64)
        out = C.c_void_p()
        check(self._L.mgta_reads_upload(self.h, packed.ctypes.data, packed.size, start_idx.ctypes.data, start_idx.size - 1,
                                        C.byref(out)), "mgta_reads_upload")
        return Reads(self, out, start_idx.size - 1)

    def adopt_reads(self, d_packed_ptr: int, n_words: int, d_start_ptr: int, n_reads: int, keepalive=None) -> "Reads":
        out = C.c_void_p()
        check(self._L.mgta_reads_adopt_device(self.h, d_packed_ptr, n_words, d_start_ptr, n_reads, C.byref(out)),
              "mgta_reads_adopt_device")
        r = Reads(self, out, n_reads)
        r._keep = keepalive
        return r

    def build_sdbg(self, reads: "Reads", k: int, min_count: int = 1, need_mercy: bool = False, collect: bool = True,
                   n_short_reads: int | None = None, bucket_range: tuple[int, int] = (0, NUM_BUCKETS)) -> EdgeStream:
        """Reads resident on the device -> edge stream (collect=False keeps it on the device: timing runs)."""
        wpt = (2 * k + 31) // 32
        recs, large, tips = [], [], []
        counts = np.zeros((NUM_BUCKETS, 3), dtype=np.int64)

        def sink(user, b0, b1, bc, r, nr, lg, nl, tp, ntw):
            nb = b1 - b0
            counts[b0:b1] = np.ctypeslib.as_array(bc, shape=(nb * 3,)).reshape(nb, 3)
            recs.append(np.ctypeslib.as_array(r, shape=(nr,)).copy() if nr and r else np.zeros(0, np.uint16))     # (NULL: keep_stream(2))
            large.append(np.ctypeslib.as_array(lg, shape=(nl,)).copy() if nl else np.zeros(0, np.uint16))
            tips.append(np.ctypeslib.as_array(tp, shape=(ntw,)).copy() if ntw and tp else np.zeros(0, np.uint32))
            return 0

        cb = _lib.EDGE_SINK(sink) if collect else C.cast(None, _lib.EDGE_SINK)
        st = _lib.BuildStats()
        ns = reads.n_reads if n_short_reads is None else n_short_reads
        check(self._L.mgta_sdbg_build_resident(self.h, reads.h, ns, k, min_count, int(need_mercy), bucket_range[0], bucket_range[1],
                                               cb, None, C.byref(st)),
              "mgta_sdbg_build_resident")
        cat = lambda xs, dt: np.concatenate(xs) if xs else np.zeros(0, dt)
        return EdgeStream(k=k, words_per_tip=wpt, bucket_items=counts[:, 0].copy(), records=cat(recs, np.uint16),
                          large=cat(large, np.uint16), tips=cat(tips, np.uint32), bucket_large=counts[:, 1].copy(),
                          bucket_tips=counts[:, 2].copy(), stats=st.as_dict())


def export_records_to_torch(ctx: "Context"):
    """records the last build left on the device (its last pass; every pass with Context.keep_stream) as a torch uint8 CUDA tensor:
    device -> device copy, no host round trip"""
    import torch
    n = C.c_uint64()
    check(ctx._L.mgta_sdbg_export_records_device(ctx.h, None, 0, C.byref(n)), "mgta_sdbg_export_records_device")
    t = torch.empty(max(1, n.value * 2), dtype=torch.uint8, device="cuda")
    check(ctx._L.mgta_sdbg_export_records_device(ctx.h, t.data_ptr(), t.numel(), C.byref(n)), "mgta_sdbg_export_records_device")
    return t[: n.value * 2]


class Graph:
    """Device-resident succinct de Bruijn graph (mgta_sdbg) <-> SuccinctDBG (succinct_dbg.h:32-247)."""

    def __init__(self, ctx: Context, stream: "EdgeStream | None", k: int = 0):
        """stream = None: the edge stream the last build of `ctx` left on the device (mgta_sdbg_load_resident; k = that build's k)"""
        if stream is None:
            self.ctx, self.k = ctx, k
            out = C.c_void_p()
            check(ctx._L.mgta_sdbg_load_resident(ctx.h, C.byref(out)), "mgta_sdbg_load_resident")
            self.h = out
            self.size = ctx._L.mgta_sdbg_size(self.h)
            return
        self.ctx, self.k = ctx, stream.k
        recs = np.ascontiguousarray(stream.records, dtype=np.uint16)
        bi = np.ascontiguousarray(stream.bucket_items, dtype=np.int64)
        tips = np.ascontiguousarray(stream.tips, dtype=np.uint32)
        out = C.c_void_p()
        check(ctx._L.mgta_sdbg_load(ctx.h, stream.k, recs.ctypes.data, recs.size, bi.ctypes.data, tips.ctypes.data, tips.size,
                                    stream.words_per_tip, C.byref(out)), "mgta_sdbg_load")
        self.h = out
        self.size = ctx._L.mgta_sdbg_size(self.h)

    @classmethod
    def from_files(cls, ctx: Context, prefix: str) -> "Graph":
        """PREFIX.sdbg_info + PREFIX.sdbg.* -> graph on the device (mgta_sdbg_load_files: the records are parsed on the device)"""
        self = cls.__new__(cls)
        self.ctx = ctx
        out = C.c_void_p()
        check(ctx._L.mgta_sdbg_load_files(ctx.h, os.fsencode(prefix), C.byref(out)), "mgta_sdbg_load_files")
        self.h = out
        self.size = ctx._L.mgta_sdbg_size(self.h)
        self.k = ctx._L.mgta_sdbg_k(self.h)
        return self

    def outgoing(self, edges) -> tuple[np.ndarray, np.ndarray]:
        """OutgoingEdges (succinct_dbg.cpp:78-97) for a batch: (outdeg int8[n], out int64[n,4])"""
        e = np.ascontiguousarray(edges, dtype=np.int64)
        out = np.empty((e.size, 4), dtype=np.int64)
        deg = np.empty(e.size, dtype=np.int8)
        check(self.ctx._L.mgta_sdbg_outgoing(self.h, e.ctypes.data, e.size, out.ctypes.data, deg.ctypes.data), "mgta_sdbg_outgoing")
        return deg, out

    def index_edges(self, kmers: list[str]) -> np.ndarray:
        """IndexBinarySearchEdge (succinct_dbg.cpp:530-549) of (k+1)-mers; -1 = absent"""
        m = {"A": 1, "C": 2, "G": 3, "T": 4, "N": 3}
        seqs = np.array([[m.get(c, 0) for c in s.upper()[: self.k + 1]] for s in kmers], dtype=np.uint8).reshape(len(kmers), self.k + 1)
        ids = np.empty(len(kmers), dtype=np.int64)
        check(self.ctx._L.mgta_sdbg_index_edges(self.h, seqs.ctypes.data, len(kmers), ids.ctypes.data), "mgta_sdbg_index_edges")
        return ids

    def invalid_bits(self) -> np.ndarray:
        """the validity bits as they are now (bit e of word e // 64 set = edge e is not part of the graph)"""
        out = np.zeros((self.size + 63) // 64, dtype=np.uint64)
        check(self.ctx._L.mgta_sdbg_invalid_bits(self.h, out.ctypes.data), "mgta_sdbg_invalid_bits")
        return out

    def denovo(self, max_tip_len: int = 150, no_bubble: bool = False, min_contig: int = 0) -> tuple[str, dict]:
        """`megagta denovo` (main_assemble, assembler.cpp:98-167): tips, bubbles, unitigs -> (text of PREFIX.contigs.fa, stats).
        The result is the reference's one-thread output.  CONSUMES the validity bits of this graph."""
        from ._lib import DenovoStats
        text, n, st = C.c_void_p(), C.c_uint64(), DenovoStats()
        check(self.ctx._L.mgta_denovo(self.h, max_tip_len, int(no_bubble), min_contig, C.byref(text), C.byref(n), C.byref(st)), "mgta_denovo")
        try:      # (ctypes.string_at takes a C int: a 100 M-read graph gives > 2^31 characters)
            fasta = bytes(memoryview((C.c_char * n.value).from_address(text.value))).decode() if n.value else ""
        finally:
            self.ctx._L.mgta_host_free(text)
        return fasta, st.as_dict()

    def free(self):
        if getattr(self, "h", None):
            self.ctx._L.mgta_sdbg_free(self.h)
            self.h = None

    __del__ = free


class DeviceHmm:
    """Profile-HMM tables on the device (mgta_hmm) <-> ProfileHMM + MostProbablePath (profile_hmm.h, most_probable_path.h)."""

    def __init__(self, ctx: Context, hm):
        self.ctx, self.M, self.A = ctx, hm.M, hm.A
        msc = np.ascontiguousarray(hm.msc, dtype=np.float64)
        tsc = np.ascontiguousarray(hm.tsc, dtype=np.float64)
        mx = np.ascontiguousarray(hm.max_match, dtype=np.float64)
        h = np.ascontiguousarray(hm.h, dtype=np.float64)
        alpha = np.ascontiguousarray(hm.alpha, dtype=np.int32)
        out = C.c_void_p()
        check(ctx._L.mgta_hmm_load(ctx.h, hm.M, hm.A, msc.ctypes.data, tsc.ctypes.data, mx.ctypes.data, h.ctypes.data, alpha.ctypes.data,
                                   C.byref(out)), "mgta_hmm_load")
        self.h = out

    def free(self):
        if getattr(self, "h", None):
            self.ctx._L.mgta_hmm_free(self.h)
            self.h = None

    __del__ = free


@dataclass
class SeedResult:
    """One seed: HMMGraphSearch::search (hmm_graph_search.h:60-81)"""
    left: str          # already reverse-complemented
    right: str
    right_side: dict
    left_side: dict

    def contig(self, kmer: str) -> str:
        return self.left + kmer.lower() + self.right


def _set_cost(ctx: "Context", cost_rate) -> None:
    """cost_rate: an int (mgta_ctx_set_search_cost_rate) or (rate, knee, rate beyond the knee) (mgta_ctx_set_search_cost_curve)"""
    if isinstance(cost_rate, (tuple, list)):
        rate, knee, rate2 = (int(x) for x in cost_rate)
        if knee:
            check(ctx._L.mgta_ctx_set_search_cost_curve(ctx.h, rate, knee, rate2), "mgta_ctx_set_search_cost_curve")
            return
        cost_rate = rate
    check(ctx._L.mgta_ctx_set_search_cost_rate(ctx.h, int(cost_rate)), "mgta_ctx_set_search_cost_rate")


def astar_search(graph: "Graph", fwd: DeviceHmm, rev: DeviceHmm, kmers: list[str], start_states, prune_len: int = 20,
                 low_cov_penalty: float = 0.5, cache_mode: int = 0, cost_rate=0) -> tuple[list[SeedResult], dict]:
    """Batched HMM-guided A* (mgta_astar_batch).  start_states[i] = model position - 1 (search.cpp:157).
    cache_mode = B >= 1 shares paths between seeds: seed j's path (c_j expansions) is seen by the seeds >= j + B + c_j // cost_rate
    (cost_rate 0: no cost term; B = 1 then is the reference's sequential run; cost_rate < 0: j + B + c_j * |cost_rate|;
    cost_rate = (rate, knee, rate2): c_j // rate up to `knee` expansions, knee // rate + (c_j - knee) // rate2 beyond).
    cache_mode = -1: no ordering at all (timing-dependent, the reference's multi-thread behaviour)."""
    ctx = graph.ctx
    _set_cost(ctx, cost_rate)
    klen = graph.k + 1
    n = len(kmers)
    for s in kmers:
        if len(s) < klen:
            raise MegaGtaError(f"seed k-mer shorter than k+1={klen}")
    buf = "".join(s[:klen] for s in kmers).encode()
    ss = np.ascontiguousarray(start_states, dtype=np.int32)
    results: list[SeedResult] = [None] * n

    def side(p):
        s = p.contents
        return dict(ok=s.ok, fval=s.fval, length=s.length, state_no=s.state_no, state=chr(s.state), partial=s.partial, node_id=s.node_id,
                    n_closed=s.n_closed, n_expanded=s.n_expanded, n_opened=s.n_opened, real_score=s.real_score, score=s.score)

    def sink(user, idx, left, ll, right, rl, rs, ls):
        results[idx] = SeedResult(left=C.string_at(left, ll).decode(), right=C.string_at(right, rl).decode(), right_side=side(rs),
                                  left_side=side(ls))
        return 0

    st = _lib.AstarStats()
    check(ctx._L.mgta_astar_batch(graph.h, fwd.h, rev.h, buf, ss.ctypes.data, n, prune_len, low_cov_penalty, cache_mode,
                                  _lib.CONTIG_SINK(sink), None, C.byref(st)), "mgta_astar_batch")
    return results, st.as_dict()


def astar_search_packed(graph: "Graph", fwd: DeviceHmm, rev: DeviceHmm, kmers: list[str], start_states, prune_len: int = 20,
                        low_cov_penalty: float = 0.5, cache_mode: int = 0, cost_rate=0, want_sides: bool = False):
    """astar_search with the results in flat arrays (mgta_astar_batch_packed): -> (contigs uint8[total], offsets int64[n + 1], stats[, sides]);
    contig i = contigs[offsets[i]:offsets[i + 1]] = left + lower-cased k-mer + right.  No Python work per seed."""
    ctx = graph.ctx
    _set_cost(ctx, cost_rate)
    klen = graph.k + 1
    n = len(kmers)
    if any(len(s) < klen for s in kmers):
        raise MegaGtaError(f"seed k-mer shorter than k+1={klen}")
    buf = "".join(s[:klen] for s in kmers).encode()
    ss = np.ascontiguousarray(start_states, dtype=np.int32)
    offsets = np.zeros(n + 1, dtype=np.uint64)
    sides = (_lib.AstarSide * (2 * n))() if want_sides else None
    text, st = C.c_void_p(), _lib.AstarStats()
    check(ctx._L.mgta_astar_batch_packed(graph.h, fwd.h, rev.h, buf, ss.ctypes.data, n, prune_len, low_cov_penalty, cache_mode, C.byref(text),
                                         offsets.ctypes.data, C.cast(sides, C.c_void_p) if want_sides else None, C.byref(st)), "mgta_astar_batch_packed")
    try:
        total = int(offsets[n])
        contigs = np.frombuffer((C.c_char * total).from_address(text.value), dtype=np.uint8).copy() if total else np.zeros(0, np.uint8)
    finally:
        ctx._L.mgta_host_free(text)
    out = (contigs, offsets.astype(np.int64), st.as_dict())
    return out + (sides,) if want_sides else out


def counting_text(hist: np.ndarray) -> str:
    """PREFIX.counting as s1_post_proc writes it: one line `i cumulative_count` for i = 1..65535 (cx1_read2sdbg_s1.cpp:923-930)"""
    acc = np.cumsum(hist[1:])
    return "".join(f"{i} {int(a)}\n" for i, a in zip(range(1, 65536), acc))


class Reads:
    def __init__(self, ctx: Context, handle, n_reads: int):
        self.ctx, self.h, self.n_reads = ctx, handle, n_reads

    def free(self):
        if getattr(self, "h", None):
            self.ctx._L.mgta_reads_free(self.h)
            self.h = None

    __del__ = free


# ------------------------------------------------------------------------------------------------
# .sdbg.N / .sdbg_info files
# ------------------------------------------------------------------------------------------------
def write_sdbg(prefix: str, s: EdgeStream, num_files: int = 1) -> None:
    """SdbgWriter layout (sdbg_multi_io.h:83-187): per record uint16, then uint16 full multiplicity if
    mult > 254, then words_per_tip uint32 if tip; text index with one line per bucket
    `bucket file byte_offset num_items num_tips num_large_mul` (file = -1 for an empty bucket).
    Buckets are dealt to `num_files` files as contiguous ranges of roughly equal record count."""
    n = s.records.size
    rec = s.records
    is_large = (rec >> 8) == 255
    is_tip = ((rec >> 5) & 1).astype(bool)
    sz = 2 + 2 * is_large.astype(np.int64) + 4 * s.words_per_tip * is_tip.astype(np.int64)
    off = np.zeros(n + 1, dtype=np.int64)
    np.cumsum(sz, out=off[1:])
    buf = np.zeros(int(off[-1]), dtype=np.uint8)
    pos = off[:-1]

    def put16(at, vals):
        v = vals.astype("<u2").view(np.uint8).reshape(-1, 2)
        buf[at] = v[:, 0]
        buf[at + 1] = v[:, 1]

    put16(pos, rec)
    if is_large.any():
        put16(pos[is_large] + 2, s.large)
    if is_tip.any():
        tp = pos[is_tip] + 2 + 2 * is_large[is_tip].astype(np.int64)
        tb = s.tips.astype("<u4").view(np.uint8).reshape(-1, 4 * s.words_per_tip)
        for j in range(4 * s.words_per_tip):
            buf[tp + j] = tb[:, j]
    bstart = np.zeros(NUM_BUCKETS + 1, dtype=np.int64)
    np.cumsum(s.bucket_items, out=bstart[1:])
    cuts = [0]
    for f in range(1, num_files):
        cuts.append(max(cuts[-1], int(np.searchsorted(bstart, n * f // num_files, side="left"))))
    cuts.append(NUM_BUCKETS)
    large_cum = np.concatenate([[0], np.cumsum(is_large)])
    tip_cum = np.concatenate([[0], np.cumsum(is_tip)])
    lines = [f"k {s.k}\n", f"words_per_tip_label {s.words_per_tip}\n", f"num_buckets {NUM_BUCKETS}\n", f"num_threads {num_files}\n",
             f"total_size {n}\n", f"num_tips {int(is_tip.sum())}\n", f"large_multi {int(is_large.sum())}\n"]
    file_of = np.zeros(NUM_BUCKETS, dtype=np.int64)
    for f in range(num_files):
        b0, b1 = cuts[f], cuts[f + 1]
        file_of[b0:b1] = f
        buf[off[bstart[b0]]:off[bstart[b1]]].tofile(f"{prefix}.sdbg.{f}")
    for b in range(NUM_BUCKETS):
        i0, i1 = bstart[b], bstart[b + 1]
        if i1 == i0:
            lines.append(f"{b} -1 0 0 0 0\n")
        else:
            f = file_of[b]
            lines.append(f"{b} {f} {off[i0] - off[bstart[cuts[f]]]} {i1 - i0} {tip_cum[i1] - tip_cum[i0]} "
                         f"{large_cum[i1] - large_cum[i0]}\n")
    with open(prefix + ".sdbg_info", "w") as fh:
        fh.writelines(lines)


def read_sdbg(prefix: str) -> EdgeStream:
    """SdbgReader (sdbg_multi_io.h:240-382): files -> logical stream in bucket order."""
    with open(prefix + ".sdbg_info") as fh:
        hdr = {}
        for key in ("k", "words_per_tip_label", "num_buckets", "num_threads", "total_size", "num_tips", "large_multi"):
            name, val = fh.readline().split()
            if name != key:
                raise ValueError(f"{prefix}.sdbg_info: expected '{key}', got '{name}'")
            hdr[key] = int(val)
        rows = np.loadtxt(fh, dtype=np.int64).reshape(-1, 6)
    if hdr["num_buckets"] != NUM_BUCKETS or rows.shape[0] != NUM_BUCKETS:
        raise ValueError("unexpected bucket count")
    wpt = hdr["words_per_tip_label"]
    files = [np.fromfile(f"{prefix}.sdbg.{t}", dtype=np.uint8) for t in range(hdr["num_threads"])]
    recs, large, tips = [], [], []
    for b in range(NUM_BUCKETS):
        _, tid, offb, items, ntips, nlarge = rows[b]
        if tid < 0 or items == 0:
            continue
        nbytes = items * 2 + nlarge * 2 + ntips * 4 * wpt
        chunk = files[tid][offb:offb + nbytes]
        if ntips == 0 and nlarge == 0:
            recs.append(chunk.view("<u2"))
            continue
        p, r, lg, tp = 0, [], [], []
        for _ in range(items):
            it = int(chunk[p]) | (int(chunk[p + 1]) << 8)
            p += 2
            r.append(it)
            if (it >> 8) == 255:
                lg.append(int(chunk[p]) | (int(chunk[p + 1]) << 8))
                p += 2
            if (it >> 5) & 1:
                tp.append(chunk[p:p + 4 * wpt].view("<u4").copy())
                p += 4 * wpt
        recs.append(np.array(r, dtype=np.uint16))
        large.append(np.array(lg, dtype=np.uint16))
        if tp:
            tips.append(np.concatenate(tp))
    cat = lambda xs, dt: np.concatenate(xs).astype(dt) if xs else np.zeros(0, dt)
    return EdgeStream(k=hdr["k"], words_per_tip=wpt, bucket_items=rows[:, 3].copy(), records=cat(recs, np.uint16),
                      large=cat(large, np.uint16), tips=cat(tips, np.uint32), bucket_large=rows[:, 5].copy(),
                      bucket_tips=rows[:, 4].copy())
