"""ctypes loader of libmegagta_hip.so (the C ABI of include/megagta_hip.h).

There is no CPU fallback: if the HIP library is missing or no device is usable every call fails
loudly (MegaGtaError).

Note: PyTorch wheels bundle their own copy of the HIP runtime.  A process that uses both this library
and torch (bench.py, megagta_amd.dist) must `import torch` BEFORE the first call into this module so
that one runtime serves both; the library itself never needs torch.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MEGAGTA_HIP_LIB") or os.path.join(_HERE, "libmegagta_hip.so")   # (override: diagnostic builds)


class MegaGtaError(RuntimeError):
    pass


class DenovoStats(C.Structure):
    _fields_ = [("n_tips", C.c_int64), ("n_bubbles", C.c_int64), ("n_bubble_candidates", C.c_int64), ("n_bubble_rounds", C.c_int64),
                ("n_paths", C.c_int64), ("n_unitig_sweeps", C.c_int64), ("n_contigs", C.c_int64), ("total_len", C.c_int64),
                ("ms_tips", C.c_float), ("ms_bubbles", C.c_float), ("ms_unitigs", C.c_float)]

    def as_dict(self):
        return {n: getattr(self, n) for n, _ in self._fields_}


class BuildStats(C.Structure):
    _fields_ = [("k", C.c_int32), ("words_per_key", C.c_int32), ("words_per_tip", C.c_int32), ("n_passes", C.c_int32),
                ("n_reads", C.c_int64), ("n_kmers", C.c_int64), ("n_items", C.c_int64), ("n_edges", C.c_int64),
                ("n_tips", C.c_int64), ("n_large", C.c_int64), ("n_sort_launches", C.c_int64), ("ms_total", C.c_double),
                ("ms_count", C.c_double), ("ms_gen", C.c_double), ("ms_sort", C.c_double), ("ms_emit", C.c_double),
                ("ms_d2h", C.c_double), ("ms_sort_scatter", C.c_double), ("ms_local_sort", C.c_double), ("n_big_segments", C.c_int64), ("n_lsd_tiles", C.c_int64),
                ("bytes_peak", C.c_uint64), ("ms_stage1", C.c_double)]

    def as_dict(self):
        return {n: getattr(self, n) for n, _ in self._fields_}


class AstarSide(C.Structure):
    _fields_ = [("ok", C.c_int32), ("fval", C.c_int32), ("length", C.c_int32), ("state_no", C.c_int32), ("state", C.c_int32),
                ("partial", C.c_int32), ("node_id", C.c_int64), ("n_closed", C.c_int64), ("n_expanded", C.c_int64),
                ("n_opened", C.c_int64), ("real_score", C.c_double), ("score", C.c_double)]


class AstarStats(C.Structure):
    _fields_ = [("n_seeds", C.c_int64), ("n_expansions", C.c_int64), ("n_opened", C.c_int64), ("n_retries", C.c_int64),
                ("ms_total", C.c_double), ("ms_kernel", C.c_double), ("n_grown", C.c_int64), ("n_rehash", C.c_int64),
                ("n_recycled", C.c_int64), ("pool_bytes", C.c_uint64), ("pool_used", C.c_uint64), ("n_resumes", C.c_int64),
                ("reserve_bytes", C.c_uint64), ("reserve_used", C.c_uint64), ("max_search_nodes", C.c_int64),
                ("max_search_expansions", C.c_int64), ("order_abandoned", C.c_int64), ("n_cache_drops", C.c_int64), ("hmm_in_lds", C.c_int64), ("n_over_limit", C.c_int64), ("ms_queue_drained", C.c_double)]

    def as_dict(self):
        return {n: getattr(self, n) for n, _ in self._fields_}


class LineProbe(C.Structure):
    _fields_ = [("waves_per_cu", C.c_int32), ("groups", C.c_int32), ("unroll", C.c_int32), ("dependent", C.c_int32), ("steps", C.c_uint64),
                ("lines_in_flight_per_cu", C.c_int32), ("pad_", C.c_int32), ("lines", C.c_uint64), ("ms", C.c_double), ("gb_per_s", C.c_double),
                ("ns_per_step", C.c_double)]

    def as_dict(self):
        return {n: getattr(self, n) for n, _ in self._fields_ if n != "pad_"}


EDGE_SINK = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int32, C.c_int32, C.POINTER(C.c_int64), C.POINTER(C.c_uint16), C.c_int64,
                        C.POINTER(C.c_uint16), C.c_int64, C.POINTER(C.c_uint32), C.c_int64)
CONTIG_SINK = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64,
                          C.POINTER(AstarSide), C.POINTER(AstarSide))

# every symbol include/megagta_hip.h declares (tests/test_abi.py checks the library exports them all)
SYMBOLS = {
    "mgta_last_error": (C.c_char_p, []),
    "mgta_version": (C.c_char_p, []),
    "mgta_ctx_create": (C.c_void_p, [C.c_int]),
    "mgta_ctx_destroy": (None, [C.c_void_p]),
    "mgta_ctx_set_mem_limit": (C.c_int, [C.c_void_p, C.c_uint64]),
    "mgta_ctx_set_full_lsd": (C.c_int, [C.c_void_p, C.c_int]),
    "mgta_sort_plan": (C.c_int, [C.c_uint64, C.c_int, C.c_uint32, C.c_uint32, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "mgta_reads_upload": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64, C.POINTER(C.c_void_p)]),
    "mgta_reads_adopt_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64, C.POINTER(C.c_void_p)]),
    "mgta_reads_free": (None, [C.c_void_p]),
    "mgta_sdbg_build_resident": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_int, C.c_int, C.c_int, C.c_int32, C.c_int32,
                                          EDGE_SINK, C.c_void_p, C.POINTER(BuildStats)]),
    "mgta_sdbg_last_counting": (C.c_int, [C.c_void_p, C.c_void_p]),
    "mgta_sdbg_export_records_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64)]),
    "mgta_sdbg_build": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64, C.c_uint64, C.c_int, C.c_int,
                                 C.c_int, EDGE_SINK, C.c_void_p, C.POINTER(BuildStats)]),
    "mgta_findstart": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]),
    "mgta_sdbg_load_resident": (C.c_int, [C.c_void_p, C.c_void_p]),
    "mgta_sdbg_load_files": (C.c_int, [C.c_void_p, C.c_char_p, C.c_void_p]),
    "mgta_sdbg_invalid_bits": (C.c_int, [C.c_void_p, C.c_void_p]),
    "mgta_ctx_set_search_cost_rate": (C.c_int, [C.c_void_p, C.c_int]),
    "mgta_ctx_set_search_cost_curve": (C.c_int, [C.c_void_p, C.c_int, C.c_uint64, C.c_int]),
    "mgta_ctx_set_search_arena": (C.c_int, [C.c_void_p, C.c_int, C.c_uint64]),
    "mgta_ctx_device_memory": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    "mgta_ctx_keep_stream": (C.c_int, [C.c_void_p, C.c_int]),
    "mgta_sdbg_stream_detach": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p)]),
    "mgta_stream_sizes": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    "mgta_stream_download": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "mgta_stream_free": (None, [C.c_void_p]),
    "mgta_sdbg_k": (C.c_int, [C.c_void_p]),
    "mgta_ctx_set_search_share": (C.c_int, [C.c_void_p, C.c_int, C.c_int]),
    "mgta_astar_batch_on": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_char_p, C.c_void_p, C.c_int64, C.c_int, C.c_double,
                                     C.c_int, CONTIG_SINK, C.c_void_p, C.POINTER(AstarStats)]),
    "mgta_astar_batch_packed": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_char_p, C.c_void_p, C.c_int64, C.c_int, C.c_double, C.c_int,
                                         C.POINTER(C.c_void_p), C.c_void_p, C.c_void_p, C.POINTER(AstarStats)]),
    "mgta_reads_pack_text": (C.c_int, [C.c_void_p, C.c_char_p, C.c_uint64, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64)]),
    "mgta_ctx_release_scratch": (C.c_int, [C.c_void_p]),
    "mgta_denovo": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    "mgta_host_free": (None, [C.c_void_p]),
    "mgta_sdbg_load": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_int64, C.c_int,
                                C.POINTER(C.c_void_p)]),
    "mgta_sdbg_free": (None, [C.c_void_p]),
    "mgta_sdbg_size": (C.c_int64, [C.c_void_p]),
    "mgta_sdbg_outgoing": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]),
    "mgta_sdbg_index_edges": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]),
    "mgta_hmm_load": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                               C.POINTER(C.c_void_p)]),
    "mgta_hmm_free": (None, [C.c_void_p]),
    "mgta_probe_random_lines": (C.c_int, [C.c_void_p, C.c_uint64, C.POINTER(LineProbe), C.c_int]),
    "mgta_astar_batch": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_char_p, C.c_void_p, C.c_int64, C.c_int, C.c_double,
                                  C.c_int, CONTIG_SINK, C.c_void_p, C.POINTER(AstarStats)]),
}

_LIB = None


def load():
    global _LIB
    if _LIB is not None:
        return _LIB
    if not os.path.exists(LIB_PATH):
        raise MegaGtaError(f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                           "(make -C megagta_amd/csrc). There is no CPU fallback.")
    L = C.CDLL(LIB_PATH)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(L, name)
        fn.restype, fn.argtypes = res, args
    _LIB = L
    return L


def check(rc: int, what: str):
    if rc != 0:
        raise MegaGtaError(f"{what} failed ({rc}): {load().mgta_last_error().decode()}")
