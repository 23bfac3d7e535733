"""ctypes binding of the CPU oracle (oracle/liboracle.so).   *** TEST INFRASTRUCTURE ***

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the
product package `megagta_amd` never does.
"""
from __future__ import annotations

import ctypes as C
import hashlib
import os
import subprocess
from dataclasses import dataclass

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build() -> str:
    subprocess.check_call(["make", "-s", "-C", _HERE, "liboracle.so"])
    return os.path.join(_HERE, "liboracle.so")


class AstarResult(C.Structure):
    _fields_ = [("ok", C.c_int32), ("fval", C.c_int32), ("length", C.c_int32), ("state_no", C.c_int32),
                ("state", C.c_int32), ("partial", C.c_int32), ("node_id", C.c_int64), ("n_closed", C.c_int64),
                ("n_expanded", C.c_int64), ("n_opened", C.c_int64), ("real_score", C.c_double), ("score", C.c_double)]


def lib():
    global _LIB
    if _LIB is not None:
        return _LIB
    path = os.path.join(_HERE, "liboracle.so")
    if not os.path.exists(path):
        build()
    L = C.CDLL(path)
    vp, i64, i32, dbl = C.c_void_p, C.c_int64, C.c_int, C.c_double
    sig = {
        "orc_sdbg_build": (vp, [vp, C.c_uint64, vp, C.c_uint64, i32, i32]),
        "orc_sdbg_build_solid": (vp, [vp, C.c_uint64, vp, C.c_uint64, C.c_uint64, i32, i32, i32, i32, vp, vp]),
        "orc_sdbg_read": (vp, [C.c_char_p]),
        "orc_stream_free": (None, [vp]),
        "orc_stream_k": (i32, [vp]), "orc_stream_words_per_tip": (i32, [vp]),
        "orc_stream_num_edges": (i64, [vp]), "orc_stream_num_tips": (i64, [vp]), "orc_stream_num_large": (i64, [vp]),
        "orc_stream_num_items_sorted": (i64, [vp]),
        "orc_stream_bucket_items": (vp, [vp]), "orc_stream_records": (vp, [vp]), "orc_stream_large": (vp, [vp]),
        "orc_stream_tips": (vp, [vp]),
        "orc_graph_from_stream": (vp, [vp]), "orc_graph_free": (None, [vp]),
        "orc_graph_size": (i64, [vp]), "orc_graph_k": (i32, [vp]), "orc_graph_f": (vp, [vp]),
        "orc_graph_w": (vp, [vp]), "orc_graph_last": (vp, [vp]), "orc_graph_tip": (vp, [vp]),
        "orc_graph_invalid": (vp, [vp]), "orc_graph_multi1": (vp, [vp]), "orc_graph_tip_labels": (vp, [vp]),
        "orc_graph_num_tips": (i64, [vp]),
        "orc_rank_last": (i64, [vp, i64]), "orc_select_last": (i64, [vp, i64]),
        "orc_rank_w": (i64, [vp, i32, i64]), "orc_select_w": (i64, [vp, i32, i64]),
        "orc_forward": (i64, [vp, i64]), "orc_backward": (i64, [vp, i64]),
        "orc_outgoing": (i32, [vp, i64, vp]), "orc_incoming": (i32, [vp, i64, vp]),
        "orc_label": (i32, [vp, i64, vp]), "orc_index_edge": (i64, [vp, vp]),
        "orc_denovo_unitig_model_check": (i64, [vp, i32, i32]), "orc_denovo_remove_tips": (i64, [vp, i32]), "orc_denovo_pop_bubbles": (i64, [vp]),
        "orc_denovo": (vp, [vp, i32, i32, i32, vp, vp, vp, vp]), "orc_free": (None, [vp]), "orc_graph_invalid_now": (vp, [vp]),
        "orc_hmm_parse": (vp, [C.c_char_p]), "orc_hmm_free": (None, [vp]),
        "orc_hmm_M": (i32, [vp]), "orc_hmm_A": (i32, [vp]),
        "orc_hmm_msc": (vp, [vp]), "orc_hmm_isc": (vp, [vp]), "orc_hmm_tsc": (vp, [vp]), "orc_hmm_maxm": (vp, [vp]),
        "orc_hmm_h": (vp, [vp]), "orc_hmm_alpha": (vp, [vp]),
        "orc_searcher_new": (vp, [vp, vp, vp, i32, dbl]), "orc_searcher_free": (None, [vp]),
        "orc_searcher_clear_cache": (None, [vp]),
        "orc_searcher_set_window": (None, [vp, i32]), "orc_searcher_set_cost_rate": (None, [vp, i32]),
        "orc_searcher_set_cost_curve": (None, [vp, i32, C.c_int64, i32]),
        "orc_search_seed": (i64, [vp, C.c_char_p, i32, C.POINTER(AstarResult), C.POINTER(AstarResult), C.c_char_p, i64]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(L, name)
        fn.restype, fn.argtypes = res, args
    _LIB = L
    return L


def _arr(ptr, n, dtype):
    if n == 0 or not ptr:
        return np.zeros(0, dtype=dtype)
    buf = (C.c_char * (int(n) * np.dtype(dtype).itemsize)).from_address(ptr)
    return np.frombuffer(buf, dtype=dtype, count=int(n)).copy()


@dataclass
class EdgeStream:
    """The logical SdBG edge stream: what `buildgraph` emits, independent of thread/file layout."""
    k: int
    words_per_tip: int
    bucket_items: np.ndarray   # int64 [65536]
    records: np.ndarray        # uint16
    large: np.ndarray          # uint16
    tips: np.ndarray           # uint32 [num_tips*words_per_tip]
    n_items_sorted: int = 0

    def md5(self) -> str:
        h = hashlib.md5()
        h.update(np.int32(self.k).tobytes())
        h.update(self.bucket_items.astype("<i8").tobytes())
        h.update(self.records.astype("<u2").tobytes())
        h.update(self.large.astype("<u2").tobytes())
        h.update(self.tips.astype("<u4").tobytes())
        return h.hexdigest()


class Stream:
    def __init__(self, handle):
        if not handle:
            raise RuntimeError("oracle: stream construction failed")
        self.h = handle

    @classmethod
    def build(cls, packed: np.ndarray, start_idx: np.ndarray, k: int, threads: int = 1) -> "Stream":
        packed = np.ascontiguousarray(packed, dtype=np.uint32)
        start_idx = np.ascontiguousarray(start_idx, dtype=np.uint64)
        return cls(lib().orc_sdbg_build(packed.ctypes.data, packed.size, start_idx.ctypes.data, start_idx.size - 1, k, threads))

    @classmethod
    def build_solid(cls, packed: np.ndarray, start_idx: np.ndarray, k: int, min_count: int, need_mercy: bool, n_short: int | None = None,
                    threads: int = 1) -> "Stream":
        """`buildgraph -m min_count [--need_mercy]`; .counting (int64[65536], per-multiplicity counts) and .n_mercy are attached"""
        packed = np.ascontiguousarray(packed, dtype=np.uint32)
        start_idx = np.ascontiguousarray(start_idx, dtype=np.uint64)
        n_reads = start_idx.size - 1
        counting = np.zeros(65536, dtype=np.int64)
        n_mercy = C.c_int64(0)
        st = cls(lib().orc_sdbg_build_solid(packed.ctypes.data, packed.size, start_idx.ctypes.data, n_reads,
                                            n_reads if n_short is None else n_short, k, min_count, int(need_mercy), threads,
                                            counting.ctypes.data, C.byref(n_mercy)))
        st.counting, st.n_mercy = counting, n_mercy.value
        return st

    @classmethod
    def read(cls, prefix: str) -> "Stream":
        return cls(lib().orc_sdbg_read(prefix.encode()))

    def edges(self) -> EdgeStream:
        L, h = lib(), self.h
        ne, nt, nl, wpt = L.orc_stream_num_edges(h), L.orc_stream_num_tips(h), L.orc_stream_num_large(h), L.orc_stream_words_per_tip(h)
        return EdgeStream(k=L.orc_stream_k(h), words_per_tip=wpt,
                          bucket_items=_arr(L.orc_stream_bucket_items(h), 65536, np.int64),
                          records=_arr(L.orc_stream_records(h), ne, np.uint16),
                          large=_arr(L.orc_stream_large(h), nl, np.uint16),
                          tips=_arr(L.orc_stream_tips(h), nt * wpt, np.uint32),
                          n_items_sorted=L.orc_stream_num_items_sorted(h))

    def __del__(self):
        if getattr(self, "h", None):
            lib().orc_stream_free(self.h)
            self.h = None


class Graph:
    def __init__(self, stream: Stream):
        self._stream = stream
        self.h = lib().orc_graph_from_stream(stream.h)
        L = lib()
        self.size = L.orc_graph_size(self.h)
        self.k = L.orc_graph_k(self.h)
        self.f = _arr(L.orc_graph_f(self.h), 6, np.int64)
        self.num_tips = L.orc_graph_num_tips(self.h)

    def bitvectors(self):
        L, h, n = lib(), self.h, self.size
        nw4, nw1 = (n + 15) // 16, (n + 63) // 64
        wpt = self._stream.edges().words_per_tip
        return dict(w=_arr(L.orc_graph_w(h), nw4, np.uint64), last=_arr(L.orc_graph_last(h), nw1, np.uint64),
                    tip=_arr(L.orc_graph_tip(h), nw1, np.uint64), invalid=_arr(L.orc_graph_invalid(h), nw1, np.uint64),
                    multi1=_arr(L.orc_graph_multi1(h), nw1, np.uint64),
                    tip_labels=_arr(L.orc_graph_tip_labels(h), self.num_tips * wpt, np.uint32))

    def outgoing(self, e: int):
        out = (C.c_int64 * 4)()
        n = lib().orc_outgoing(self.h, e, out)
        return n, [out[i] for i in range(max(n, 0))]

    def incoming(self, e: int):
        out = (C.c_int64 * 4)()
        n = lib().orc_incoming(self.h, e, out)
        return n, [out[i] for i in range(max(n, 0))]

    def rank_last(self, p): return lib().orc_rank_last(self.h, p)
    def select_last(self, r): return lib().orc_select_last(self.h, r)
    def rank_w(self, c, p): return lib().orc_rank_w(self.h, c, p)
    def select_w(self, c, r): return lib().orc_select_w(self.h, c, r)
    def forward(self, e): return lib().orc_forward(self.h, e)
    def backward(self, e): return lib().orc_backward(self.h, e)

    def label(self, e: int) -> str:
        buf = (C.c_uint8 * (self.k + 2))()
        lib().orc_label(self.h, e, buf)
        return "".join("$ACGT"[buf[i]] for i in range(self.k))

    def index_edge(self, kmer: str) -> int:
        m = {"A": 1, "C": 2, "G": 3, "T": 4, "N": 3}
        buf = (C.c_uint8 * (self.k + 2))(*[m[c] for c in kmer.upper()[: self.k + 1]])
        return lib().orc_index_edge(self.h, buf)

    def denovo(self, max_tip_len: int = 150, no_bubble: bool = False, min_contig: int = 0):
        """`megagta denovo` as the reference's one-thread run; CONSUMES the validity bits of this graph.
        Returns (fasta_text, dict(n_contigs, total_len, n_tips, n_bubbles))."""
        n = (C.c_int64 * 4)()
        ptr = lib().orc_denovo(self.h, max_tip_len, int(no_bubble), min_contig, C.byref(n, 0), C.byref(n, 8), C.byref(n, 16), C.byref(n, 24))
        text = C.string_at(ptr).decode()
        lib().orc_free(ptr)
        return text, dict(n_contigs=n[0], total_len=n[1], n_tips=n[2], n_bubbles=n[3])

    def invalid_now(self) -> np.ndarray:
        return _arr(lib().orc_graph_invalid_now(self.h), (self.size + 63) // 64, np.uint64).copy()

    def __del__(self):
        if getattr(self, "h", None):
            lib().orc_graph_free(self.h)
            self.h = None


class Hmm:
    def __init__(self, path: str):
        self.h = lib().orc_hmm_parse(path.encode())
        if not self.h:
            raise FileNotFoundError(path)
        L = lib()
        self.M, self.A = L.orc_hmm_M(self.h), L.orc_hmm_A(self.h)
        M, A = self.M, self.A
        self.msc = _arr(L.orc_hmm_msc(self.h), (M + 1) * A, np.float64).reshape(M + 1, A)
        self.isc = _arr(L.orc_hmm_isc(self.h), (M + 1) * A, np.float64).reshape(M + 1, A)
        self.tsc = _arr(L.orc_hmm_tsc(self.h), 7 * (M + 1), np.float64).reshape(7, M + 1)
        self.maxm = _arr(L.orc_hmm_maxm(self.h), M + 1, np.float64)
        self.hcost = _arr(L.orc_hmm_h(self.h), 3 * (M + 1), np.float64).reshape(3, M + 1)
        self.alpha = _arr(L.orc_hmm_alpha(self.h), 127, np.int32)

    def __del__(self):
        if getattr(self, "h", None):
            lib().orc_hmm_free(self.h)
            self.h = None


class Searcher:
    def __init__(self, graph: Graph, fwd: Hmm, rev: Hmm, prune_len: int = 20, low_cov_penalty: float = 0.5):
        self._keep = (graph, fwd, rev)
        self.h = lib().orc_searcher_new(graph.h, fwd.h, rev.h, prune_len, low_cov_penalty)

    def clear_cache(self):
        lib().orc_searcher_clear_cache(self.h)

    def set_cost_rate(self, rate):
        """the path of seed j (c_j expansions) is seen by the seeds >= j + window + c_j // rate (0 = no cost term);
        (rate, knee, rate2): c_j // rate up to `knee` expansions, knee // rate + (c_j - knee) // rate2 beyond"""
        if isinstance(rate, (tuple, list)) and rate[1]:
            lib().orc_searcher_set_cost_curve(self.h, int(rate[0]), int(rate[1]), int(rate[2]))
        else:
            lib().orc_searcher_set_cost_rate(self.h, int(rate[0] if isinstance(rate, (tuple, list)) else rate))

    def set_window(self, window: int):
        """seed j sees the paths of seeds <= j - window (1 = sequential sharing like `search ... 1`)"""
        lib().orc_searcher_set_window(self.h, window)

    def search(self, kmer: str, start_state: int, cold: bool = True):
        """Returns (contig, right_result, left_result). cold=True drops the term_nodes caches first."""
        if cold:
            self.clear_cache()
        r, l = AstarResult(), AstarResult()
        buf = C.create_string_buffer(1 << 16)
        n = lib().orc_search_seed(self.h, kmer.encode(), start_state, C.byref(r), C.byref(l), buf, len(buf))
        if n < 0:
            raise RuntimeError(f"oracle search failed ({n})")
        return buf.value.decode(), r, l

    def __del__(self):
        if getattr(self, "h", None):
            lib().orc_searcher_free(self.h)
            self.h = None
