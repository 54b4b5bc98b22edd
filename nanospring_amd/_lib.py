"""ctypes binding of libnsgpu.so.  Signatures follow include/nsgpu.h one to one."""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))


class NsGpuError(RuntimeError):
    """Mirrors the std::runtime_error the reference throws (src/main.cpp:161-176)."""


def lib_path():
    return os.path.join(HERE, "lib", "libnsgpu.so")


class Params(C.Structure):
    _fields_ = [("k", C.c_uint32), ("n", C.c_uint32), ("overlap_sketch_thr", C.c_uint32), ("m_k", C.c_uint32),
                ("m_w", C.c_uint32), ("max_chain_iter", C.c_uint32), ("edge_threshold", C.c_uint64),
                ("device", C.c_int32), ("reserved", C.c_int32)]


class Timing(C.Structure):
    _fields_ = [("pack_ms", C.c_float), ("sketch_ms", C.c_float), ("index_ms", C.c_float), ("filter_ms", C.c_float),
                ("repetitive_ms", C.c_float), ("sketch_kernel_ms", C.c_float), ("filter_kernel_ms", C.c_float),
                ("filter_matches", C.c_uint64)]


_u8p, _u32p, _u64p = C.POINTER(C.c_uint8), C.POINTER(C.c_uint32), C.POINTER(C.c_uint64)
_vp = C.c_void_p

# name -> (restype, argtypes); the CPU test suite checks every name is exported.
SIGNATURES = {
    "nsgpu_default_params": (None, [C.POINTER(Params)]),
    "nsgpu_create": (C.c_int, [C.POINTER(Params), C.POINTER(_vp)]),
    "nsgpu_destroy": (None, [_vp]),
    "nsgpu_free": (None, [_vp]),
    "nsgpu_last_error": (C.c_char_p, []),
    "nsgpu_set_stream": (C.c_int, [_vp, _vp]),
    "nsgpu_sync": (C.c_int, [_vp]),
    "nsgpu_version": (C.c_char_p, []),
    "nsgpu_load_reads_ascii": (C.c_int, [_vp, _vp, _vp, C.c_uint32]),
    "nsgpu_load_reads_packed": (C.c_int, [_vp, _vp, _vp, _vp, C.c_uint32]),
    "nsgpu_num_reads": (C.c_uint32, [_vp]),
    "nsgpu_num_bases": (C.c_uint64, [_vp]),
    "nsgpu_get_read": (C.c_int, [_vp, C.c_uint32, _vp, _u32p]),
    "nsgpu_get_read_packed": (C.c_int, [_vp, C.c_uint32, _vp, _u32p]),
    "nsgpu_sketch": (C.c_int, [_vp, _vp, _vp]),
    "nsgpu_build_index": (C.c_int, [_vp]),
    "nsgpu_index_export": (C.c_int, [_vp, C.c_uint32, _vp, _vp, _vp, _u32p]),
    "nsgpu_filter": (C.c_int, [_vp, _vp, C.c_size_t, C.POINTER(_vp), C.POINTER(C.c_size_t)]),
    "nsgpu_filter_batch": (C.c_int, [_vp, _vp, _vp, C.c_uint32, C.POINTER(_vp), C.POINTER(_vp)]),
    "nsgpu_filter_all_reads": (C.c_int, [_vp, _u64p]),
    "nsgpu_filter_all_fetch": (C.c_int, [_vp, _vp, _vp]),
    "nsgpu_check_repetitive": (C.c_int, [_vp, _vp]),
    "nsgpu_ksw_extd2_batch": (C.c_int, [_vp, C.c_uint32] + [_vp] * 11 + [C.POINTER(_vp), C.POINTER(_vp)]),
    "nsgpu_load_fastq": (C.c_int, [_vp, _vp, C.c_size_t, _u32p]),
    "nsgpu_mm_sketch_batch": (C.c_int, [_vp, _vp, _vp, C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(_vp), C.POINTER(_vp)]),
    "nsgpu_chain_scores": (C.c_int, [_vp, _vp, _vp, C.c_uint32, _vp, _vp]),
    "nsgpu_seed_anchors": (C.c_int, [_vp, _vp, _vp, C.c_uint32, _vp, _vp, _vp, C.c_uint32, C.POINTER(_vp), C.POINTER(_vp), _vp, _vp, _vp]),
    "nsgpu_align_batch": (C.c_int, [_vp, _vp, _vp, C.c_uint32, _vp, _vp, _vp, C.c_uint32, _vp, C.POINTER(_vp), C.POINTER(_vp)]),
    "nsgpu_get_align_stats": (C.c_int, [_vp, _vp]),
    "nsgpu_host_wait_count": (C.c_uint64, []),
    "nsgpu_bwt_block": (C.c_int, [_vp, _vp, C.c_uint64, _vp, _vp, C.c_uint32, _vp, _vp, _vp, _vp]),
    "nsgpu_reset_align_stats": (C.c_int, [_vp]),
    "nsgpu_set_schedule": (C.c_int, [_vp, C.c_uint32, C.c_uint32, C.c_uint32]),
    "nsgpu_set_schedule2": (C.c_int, [_vp, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32]),
    "nsgpu_set_schedule_auto": (C.c_int, [_vp]),
    "nsgpu_set_defer": (C.c_int, [_vp, C.c_uint32, C.c_uint32]),
    "nsgpu_get_defer": (C.c_int, [_vp, _u32p, _u32p, C.POINTER(C.c_uint64)]),
    "nsgpu_set_graph": (C.c_int, [_vp, C.c_uint32]),
    "nsgpu_get_graph_stats": (C.c_int, [_vp, _vp]),
    "nsgpu_get_schedule2": (C.c_int, [_vp, _u32p, _u32p, _u32p, _u32p, _u32p]),
    "nsgpu_get_schedule": (C.c_int, [_vp, _u32p, _u32p, _u32p]),
    "nsgpu_cons_begin": (C.c_int, [_vp, C.c_uint32, C.c_uint32, C.c_uint32]),
    "nsgpu_cons_groups": (C.c_uint32, []),
    "nsgpu_cons_slot": (C.c_int, [_vp, C.c_uint32]),
    "nsgpu_cons_advance": (C.c_int, [_vp, C.c_int, C.c_int]),
    "nsgpu_cons_seed_requests": (C.c_int, [_vp, C.c_int, C.POINTER(_vp), C.POINTER(_vp), _u32p]),
    "nsgpu_cons_seed_resolve": (C.c_int, [_vp, _vp, _vp, C.c_uint32, _u32p, _u32p]),
    "nsgpu_cons_batches": (C.c_int, [_vp, C.c_int]),
    "nsgpu_cons_claim_requests": (C.c_int, [_vp, C.c_int, C.POINTER(_vp), C.POINTER(_vp), _u32p]),
    "nsgpu_cons_claim_resolve": (C.c_int, [_vp, _vp, _vp, C.c_uint32, _u32p]),
    "nsgpu_cons_finish": (C.c_int, [_vp, C.c_uint32, _vp]),
    "nsgpu_load_fastq_begin": (C.c_int, [_vp]),
    "nsgpu_load_fastq_chunk": (C.c_int, [_vp, _vp, C.c_size_t]),
    "nsgpu_load_fastq_end": (C.c_int, [_vp, _u32p]),
    "nsgpu_load_fastq_file": (C.c_int, [_vp, C.c_char_p, C.c_int, _u32p]),
    "nsgpu_comm_unique_id": (C.c_int, [_vp]),
    "nsgpu_comm_init_rccl": (C.c_int, [_vp, _vp, C.c_uint32, C.c_uint32, C.POINTER(_vp)]),
    "nsgpu_comm_init_callbacks": (C.c_int, [_vp, _vp, C.c_uint32, C.c_uint32, C.POINTER(_vp)]),
    "nsgpu_comm_destroy": (None, [_vp]),
    "nsgpu_comm_stats": (C.c_int, [_vp, _vp, _vp, _vp]),
    "nsgpu_dist_load_reads": (C.c_int, [_vp, _vp, _vp, _vp, C.c_uint32, _u32p, _u32p]),
    "nsgpu_dist_sketch_index": (C.c_int, [_vp, _vp, _vp, C.c_int]),
    "nsgpu_dist_consensus_run": (C.c_int, [_vp, _vp, C.c_uint32, C.c_uint32, _vp]),
    "nsgpu_sketch_range": (C.c_int, [_vp, _vp, C.c_uint32, C.c_uint32]),
    "nsgpu_sketch_rows_get": (C.c_int, [_vp, C.c_uint32, C.c_uint32, _vp, C.c_int]),
    "nsgpu_sketch_rows_set": (C.c_int, [_vp, C.c_uint32, C.c_uint32, _vp, C.c_int]),
    "nsgpu_sketch_mark_complete": (C.c_int, [_vp]),
    "nsgpu_set_read_id_base": (C.c_int, [_vp, C.c_uint32]),
    "nsgpu_consensus_run": (C.c_int, [_vp, C.c_uint32, C.c_uint32, _vp]),
    "nsgpu_consensus_stream": (C.c_int, [_vp, C.c_uint32, C.c_uint32, C.POINTER(_vp), C.POINTER(C.c_size_t)]),
    "nsgpu_consensus_write": (C.c_int, [_vp, C.c_char_p, C.c_char_p]),
    "nsgpu_consensus_verify": (C.c_int, [_vp, _u64p]),
    "nsgpu_get_timing": (C.c_int, [_vp, C.POINTER(Timing)]),
    "nsgpu_synth_reads": (C.c_int, [C.c_uint64, C.c_uint64, C.c_uint32, C.c_double, C.c_double, C.c_double, C.c_double,
                                    C.POINTER(_vp), C.POINTER(_vp)]),
    "nsgpu_synth_reads_kind": (C.c_int, [C.c_uint64, C.c_uint64, C.c_uint32, C.c_uint32, C.c_double, C.c_double, C.c_double, C.c_double, C.c_uint32,
                                         C.POINTER(C.c_void_p), C.POINTER(C.c_void_p)]),
    "nsgpu_synth_reads_range": (C.c_int, [C.c_uint64, C.c_uint64, C.c_uint32, C.c_uint32, C.c_double, C.c_double, C.c_double, C.c_double,
                                          C.POINTER(_vp), C.POINTER(_vp)]),
}

_LIB = None


def load_library():
    """Loads libnsgpu.so or raises NsGpuError -- never substitutes anything else."""
    global _LIB
    if _LIB is not None:
        return _LIB
    p = lib_path()
    if not os.path.exists(p):
        raise NsGpuError(f"{p} is missing: run `python -m nanospring_amd.build` (there is no CPU fallback)")
    try:
        lib = C.CDLL(p)
    except OSError as e:  # pragma: no cover
        raise NsGpuError(f"cannot load {p}: {e}") from e
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise NsGpuError(f"{p} does not export {name}") from e
        fn.restype = res
        fn.argtypes = args
    _LIB = lib
    return lib


def check(lib, rc):
    if rc != 0:
        msg = lib.nsgpu_last_error()
        raise NsGpuError(f"nsgpu error {rc}: {msg.decode() if msg else '?'}")
