"""ctypes binding of libidelucs_hip.so (the C ABI in include/idelucs_hip.h).

There is no CPU fallback anywhere in this package: if the shared library is missing the import
fails, and if no gfx950 device is usable every compute entry point raises RuntimeError.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# IDELUCS_DEV="name=value,name=value": developer / diagnostic settings for the variant tables of fused.py, posthoc.py and utils.py (VARIANTS / OPTIONS:
# launch-sequence and algorithm variants that were measured and not adopted, kept for the tests and tools that compare them with the default)
DEV = dict(kv.split("=", 1) for kv in os.environ.get("IDELUCS_DEV", "").split(",") if "=" in kv)
LIB_PATH = os.environ.get("IDELUCS_LIB_PATH") or os.path.join(_HERE, "csrc", "libidelucs_hip.so")   # (override: A/B of two builds)

IDL_FALLBACK = 1
IDL_OK, IDL_ERR_ARG, IDL_ERR_HIP, IDL_ERR_IO, IDL_ERR_HEADER, IDL_ERR_BASE, IDL_ERR_NOMEM = 0, -1, -2, -3, -4, -5, -6
MODE_KMER, MODE_CGR, MODE_CANONICAL = 0, 1, 2
INIT_ZERO, INIT_ONE, INIT_FROM_OUT = 0, 1, 2
OUT_COUNTS_I32, OUT_FREQ_F32, OUT_FREQ_F64 = 0, 1, 2
MAX_K = 7

_c = ctypes
_vp, _i64, _i32, _int = _c.c_void_p, _c.c_int64, _c.c_int32, _c.c_int
_pi64, _pu8 = _c.POINTER(_c.c_int64), _c.POINTER(_c.c_uint8)

# name -> (restype, argtypes); kept in one table so tests can check every symbol the header declares
SIGNATURES = {
    "idl_last_error": (_c.c_char_p, []),
    "idl_abi_version": (_int, []),
    "idl_device_count": (_int, []),
    "idl_row_len": (_i64, [_int, _int]),
    "idl_kmer_counts": (_int, [_vp, _i64, _int, _vp]),
    "idl_cgr": (_int, [_vp, _i64, _int, _vp]),
    "idl_kmer_rev_comp": (_int, [_vp, _int, _vp]),
    "idl_check_sequence": (_int, [_vp, _i64, _vp, _pi64, _pi64]),
    "idl_pack": (_int, [_vp, _vp, _i64, _vp, _vp, _vp]),
    "idl_fasta_open": (_int, [_c.c_char_p, _int, _c.POINTER(_vp)]),
    "idl_fasta_close": (None, [_vp]),
    "idl_fasta_sizes": (_int, [_vp, _pi64, _pi64, _pi64, _pi64]),
    "idl_fasta_export": (_int, [_vp] + [_vp] * 8),
    "idl_fasta_pack_range": (_int, [_vp, _i64, _i64, _vp, _vp]),
    "idl_ingest_threads": (_int, []),
    "idl_ingest_numa_node": (_int, []),
    "idl_ingest_file_node": (_int, []),
    "idl_ingest_probe_file_node": (_int, [_c.c_char_p]),
    "idl_ingest_cpu_plan": (_int, [_int, _int, _vp, _vp]),
    "idl_fasta_parse_pack": (_int, [_c.c_char_p, _vp, _vp, _i64, _vp, _vp, _vp, _c.POINTER(_vp)]),
    "idl_fasta_arena_slots": (_int, [_vp, _vp]),
    "idl_fasta_arena_meta": (_int, [_vp, _vp, _vp, _pi64, _pi64]),
    "idl_fasta_names_high": (_int, [_vp]),
    "idl_fasta_arena_mask_flags": (_i64, [_vp, _vp]),
    "idl_mask_from_lengths": (_int, [_vp, _vp, _vp, _vp, _i64, _vp]),
    "idl_ingest_release": (None, []),
    "idl_vectorise": (_int, [_vp, _vp, _vp, _vp, _i64, _int, _int, _int, _int, _int, _vp, _vp, _vp, _i64, _i64, _vp]),
    "idl_mimic_workspace": (_i64, [_i64, _int]),
    "idl_mimic_max_random_n": (_int, []),
    "idl_mimic_edits": (_int, [_vp, _i64, _int, _vp, _vp, _vp, _c.c_uint64, _vp, _vp, _i64, _pi64, _vp, _vp]),
    "idl_col_stats_workspace": (_i64, [_i64, _i64]),
    "idl_col_stats": (_int, [_vp, _int, _i64, _i64, _vp, _vp, _vp, _vp]),
    "idl_standardise": (_int, [_vp, _int, _i64, _i64, _vp, _vp, _vp, _vp]),
    "idl_counts_stats_workspace": (_i64, [_i64, _i64]),
    "idl_counts_stats": (_int, [_vp, _i64, _i64, _vp, _vp, _vp, _vp, _vp]),
    "idl_counts_standardise": (_int, [_vp, _vp, _i64, _i64, _vp, _vp, _vp, _vp]),
    "idl_gather_pairs": (_int, [_vp, _i64, _i64, _i64, _vp, _i64, _vp, _vp, _vp, _vp]),
    "idl_gather_pairs_at": (_int, [_vp, _i64, _i64, _i64, _vp, _vp, _i64, _vp, _vp, _vp, _vp, _vp]),
    "idl_relu_dropout_fwd": (_int, [_vp, _i64, _int, _c.c_uint64, _vp, _int, _vp]),
    "idl_head_fwd": (_int, [_vp, _vp, _vp, _int, _int, _int, _c.c_uint64, _vp, _vp, _vp, _vp, _vp, _vp]),
    "idl_mid_fwd": (_int, [_vp, _vp, _vp, _vp, _vp, _int, _int, _int, _c.c_uint64, _vp, _vp, _vp, _vp, _vp, _vp]),
    "idl_nce_rows": (_int, [_vp, _int, _c.c_float, _vp, _vp, _vp]),
    "idl_iic_core": (_int, [_vp, _int, _c.c_float, _c.c_float, _c.c_float, _vp, _vp, _vp]),
    "idl_iic_core_dz": (_int, [_vp, _int, _c.c_float, _c.c_float, _c.c_float, _vp, _vp, _vp, _int, _vp, _vp]),
    "idl_head_bwd": (_int, [_vp] * 5 + [_int, _vp, _vp, _int, _int, _int, _c.c_float, _vp, _vp, _vp]),
    "idl_head_bwd_dz": (_int, [_vp] * 5 + [_int, _vp, _vp, _int, _int, _int, _c.c_float, _vp, _vp, _vp]),
    "idl_nce_fused_workspace": (_i64, [_int]),
    "idl_nce_fused_parts": (_int, []),
    "idl_nce_fused": (_int, [_vp, _int, _c.c_float, _vp, _vp, _vp, _vp, _vp]),
    "idl_nce_fused_iic": (_int, [_vp, _int, _c.c_float, _vp, _vp, _vp, _vp, _vp, _int, _c.c_float, _c.c_float, _c.c_float, _vp, _vp, _vp]),
    "idl_nce_fused_iic_z": (_int, [_vp, _int, _c.c_float, _vp, _vp, _vp, _vp, _vp, _vp, _int, _c.c_float, _c.c_float, _c.c_float, _vp, _vp, _vp]),
    "idl_col_sum": (_int, [_vp, _int, _int, _vp, _vp]),
    "idl_relu_dropout_bwd_colsum": (_int, [_vp, _vp, _int, _int, _int, _vp, _vp]),
    "idl_col_sum_parts": (_int, []),
    "idl_mid_bwd": (_int, [_vp] * 5 + [_int] + [_vp] * 4 + [_int, _int, _int, _c.c_float] + [_vp] * 8 + [_i64, _vp]),
    "idl_bias_grads": (_int, [_vp, _vp, _int, _vp, _vp, _int, _vp, _vp, _int, _vp, _int, _int, _vp, _i64, _vp, _vp, _vp]),
    "idl_rmsprop_step_gather": (_int, [_int, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _int, _c.c_float, _c.c_float, _vp,
                                       _vp, _i64, _i64, _i64, _vp, _i64, _i64, _vp, _vp, _vp, _vp, _vp]),
    "idl_wgrad_supported": (_int, [_int, _int, _int]),
    "idl_wgrad_rmsprop": (_int, [_vp, _vp, _int, _int, _int, _vp, _vp, _vp, _vp, _vp]),
    "idl_l1_fwd_supported": (_int, [_int, _int, _int]),
    "idl_l1_fwd": (_int, [_vp, _vp, _int, _int, _vp, _vp]),
    "idl_debug_wgrad_clock": (_int, [_vp, _vp, _int, _int, _int, _vp, _int, _vp, _vp]),
    "idl_rmsprop_step_gather_wgrad": (_int, [_int, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _int, _c.c_float, _c.c_float, _vp,
                                             _vp, _i64, _i64, _i64, _vp, _i64, _i64, _vp, _vp, _vp, _vp,
                                             _int, _vp, _vp, _int, _int, _int, _int, _vp, _i64, _vp]),
    "idl_wgrad_rmsprop_step": (_int, [_int, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _int, _c.c_float, _c.c_float, _vp,
                                      _int, _vp, _vp, _int, _int, _int, _vp,
                                      _int, _vp, _vp, _int, _int, _int, _int, _vp, _i64, _vp]),
    "idl_l1_fwd_rms": (_int, [_vp, _vp, _int, _int, _vp,
                              _int, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _int, _c.c_float, _c.c_float, _vp, _int,
                              _int, _vp, _vp, _int, _int, _int, _int, _vp, _i64, _vp]),
    "idl_mid_bwd_gather": (_int, [_vp] * 5 + [_int] + [_vp] * 4 + [_int, _int, _int, _c.c_float] + [_vp] * 7 +
                           [_vp, _i64, _i64, _i64, _vp, _vp, _i64, _i64, _i64, _vp, _vp, _vp, _vp, _int, _int, _int, _int, _vp]),
    "idl_mid_fwd_gather": (_int, [_vp, _vp, _int, _vp, _vp, _vp, _vp, _int, _int, _int, _c.c_uint64, _vp, _vp, _vp, _vp, _vp] +
                           [_vp, _i64, _i64, _i64, _vp, _vp, _i64, _i64, _i64, _vp, _vp, _vp, _vp, _int, _int, _int, _vp]),
    "idl_planes_exponent": (_int, [_int]),
    "idl_split_planes": (_int, [_vp, _i64, _int, _vp, _vp, _vp, _vp]),
    "idl_l1_planes_parts": (_int, []),
    "idl_l1_planes_supported": (_int, [_int, _int, _int]),
    "idl_l1_planes": (_int, [_vp, _vp, _int, _vp, _vp, _int, _int, _int, _int, _vp, _vp]),
    "idl_reduce_parts_rms": (_int, [_vp, _i64, _vp, _vp,
                                    _int, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _int, _c.c_float, _c.c_float, _vp, _int,
                                    _int, _vp, _vp, _int, _int, _int, _int, _vp, _i64, _vp]),
    "idl_wgrad_xplanes_supported": (_int, [_int, _int, _int]),
    "idl_wgrad_rmsprop_xplanes": (_int, [_vp, _vp, _vp, _vp, _vp, _int, _int, _int, _int, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "idl_wgrad_xplanes_rms": (_int, [_vp, _vp, _vp, _vp, _vp, _int, _int, _int, _int, _vp, _vp, _vp, _vp, _vp, _vp,
                                     _int, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _int, _c.c_float, _c.c_float, _vp, _int,
                                     _int, _vp, _vp, _int, _int, _int, _int, _vp, _i64, _vp]),
    "idl_wgrad_rmsprop_planes": (_int, [_vp, _vp, _int, _int, _int, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "idl_mid_bwd_gather_planes": (_int, [_vp] * 5 + [_int] + [_vp] * 4 + [_int, _int, _int, _c.c_float] + [_vp] * 7 +
                                  [_vp, _i64, _i64, _i64, _vp, _vp, _i64, _i64, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _int, _int, _int, _int,
                                   _vp, _vp, _vp, _vp, _vp]),
    "idl_dr1_scale_words": (_int, []),
    "idl_mid_fwd_gather_planes": (_int, [_vp, _vp, _int, _vp, _vp, _vp, _vp, _int, _int, _int, _c.c_uint64, _vp, _vp, _vp, _vp, _vp] +
                                  [_vp, _i64, _i64, _i64, _vp, _vp, _i64, _i64, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _int, _int, _int, _vp]),
    "idl_mst_prim_workspace": (_i64, [_i64]),
    "idl_mst_prim": (_int, [_vp, _int, _vp, _i64, _int, _vp, _vp, _vp, _vp, _vp]),
    "idl_mst_prim_local": (_int, [_vp, _int, _vp, _i64, _int, _vp, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "idl_debug_prim_phases": (_int, [_vp]),
    "idl_debug_lazy_phases": (_int, [_vp]),
    "idl_mst_prim_lazy_workspace": (_i64, [_i64, _int]),
    "idl_mst_prim_lazy": (_int, [_vp, _vp, _vp, _i64, _int, _vp, _i64, _vp, _vp, _vp, _vp, _vp, _int, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "idl_silhouette_sums": (_int, [_vp, _vp, _vp, _i64, _int, _int, _vp, _vp]),
    "idl_nce_fused_joint": (_int, [_vp, _int, _c.c_float, _vp, _vp, _vp, _vp, _vp, _vp, _int, _vp]),
    "idl_iic_joint": (_int, [_vp, _int, _int, _vp, _vp]),
    "idl_at_b": (_int, [_vp, _int, _vp, _int, _int, _int, _int, _vp, _int, _vp]),
    "idl_knn_window": (_int, [_vp, _i64, _int, _vp, _vp, _i64, _i64, _vp, _vp, _vp, _vp, _vp, _int, _vp]),
    "idl_knn_select": (_int, [_vp, _i64, _int, _vp, _vp, _vp, _i64, _i64, _i64, _vp, _vp, _vp, _vp, _int, _vp, _vp, _vp]),
    "idl_debug_stamps": (_int, [_vp]),
    "idl_debug_phase_stamps": (_int, [_int]),
    "idl_mimic_check_lengths": (_int, [_i64, _int, _vp, _vp]),
    "idl_mimic_slots_workspace": (_i64, [_int]),
    "idl_mimic_slots_capacity": (_i64, [_i64, _int, _vp, _vp, _vp, _i64]),
    "idl_mimic_edits_slots": (_int, [_vp, _i64, _int, _vp, _vp, _vp, _c.c_uint64, _i64, _vp, _vp, _i64, _vp, _vp, _vp]),
    "idl_vectorise_ranges": (_int, [_vp, _vp, _vp, _vp, _i64, _int, _int, _int, _int, _int, _vp, _vp, _vp, _i64, _i64, _vp]),
    "idl_plan_bytes": (_i64, []),
    "idl_plan_begin": (_int, [_vp]),
    "idl_plan_end": (_int, []),
    "idl_plan_launch": (_int, [_vp, _vp, _int, _vp]),
    "idl_rmsprop_step": (_int, [_int, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _vp, _int, _c.c_float, _c.c_float, _vp, _vp]),
}

if not os.path.exists(LIB_PATH):
    raise ImportError(
        f"{LIB_PATH} is missing: build it with `make -C idelucs_amd/csrc` (or __graft_entry__.build()). "
        "idelucs_amd has no CPU fallback.")

lib = ctypes.CDLL(LIB_PATH)
for _name, (_res, _args) in SIGNATURES.items():
    _fn = getattr(lib, _name)
    _fn.restype = _res
    _fn.argtypes = _args


def last_error():
    msg = lib.idl_last_error()
    return msg.decode("utf-8", "replace") if msg else ""


def check(rc):
    """Map a C-ABI return code onto the exception the reference raises for the same condition."""
    if rc == IDL_OK:
        return
    msg = last_error()
    if rc in (IDL_ERR_HEADER, IDL_ERR_BASE, IDL_ERR_ARG):
        raise ValueError(msg)              # reference: ValueError (utils.py:38,40,50)
    if rc == IDL_ERR_IO:
        raise FileNotFoundError(msg)       # reference: open() failure
    if rc == IDL_ERR_NOMEM:
        raise MemoryError(msg)
    raise RuntimeError(msg or f"libidelucs_hip error {rc}")


def require_gpu():
    if lib.idl_device_count() < 1:
        raise RuntimeError("idelucs_amd needs a gfx950 (MI355X) GPU: no usable HIP device found, "
                           "and there is no CPU fallback")
