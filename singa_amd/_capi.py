"""ctypes binding of the C ABI in include/singa_hip.h (one place, so that the GPU library and the test-only CPU
emulation build are called through identical signatures)."""
import ctypes as C

P = C.c_void_p
I32 = C.c_int
I64 = C.c_int64
F32 = C.c_float


class Seg(C.Structure):
    _fields_ = [("ptr", P), ("ld", I64), ("rows", C.c_int32)]


SegArr = Seg * 3


class Gemm(C.Structure):
    """singa_gemm_t of include/singa_hip.h"""
    _fields_ = [("a", P), ("b", P), ("c", P), ("bias", P), ("lda", I64), ("ldb", I64), ("ldc", I64), ("I", C.c_int32),
                ("J", C.c_int32), ("R", C.c_int32), ("a_group", C.c_int32), ("b_group", C.c_int32), ("c_group", C.c_int32),
                ("a_group_ld", I64), ("b_group_ld", I64), ("c_group_ld", I64), ("c_split_stride", I64), ("mask", P),
                ("addend", P), ("relu", C.c_int32), ("asum", P), ("asum_stride", I64)]


class CGemm(C.Structure):
    """singa_cgemm_t of include/singa_hip.h"""
    _fields_ = [("a", P), ("b", P), ("c", P), ("lda", I64), ("ldb", I64), ("ldc", I64), ("a_im", I64), ("b_im", I64),
                ("c_im", I64), ("c_split_stride", I64), ("I", C.c_int32), ("J", C.c_int32), ("R", C.c_int32),
                ("sigma", C.c_float)]


def cgemm_probs(items):
    arr = (CGemm * len(items))()
    for g, it in zip(arr, items):
        for k, v in it.items():
            setattr(g, k, v)
    return arr, len(items)


def gemm_probs(items):
    """items: list of dicts with the fields of singa_gemm_t (pointers as ints, missing fields 0) -> (ctypes array, n)"""
    arr = (Gemm * len(items))()
    for g, it in zip(arr, items):
        for k, v in it.items():
            setattr(g, k, v)
    return arr, len(items)

_SIGS = {
    "singa_version": ([], I32),
    "singa_last_error_string": ([], C.c_char_p),
    "singa_init": ([P, I32], I32),
    "singa_dims": ([I32, I32, C.POINTER(I32), C.POINTER(I32), C.POINTER(I32)], I32),
    "singa_edge_frames": ([P, P, P, P, I32, P], I32),
    "singa_wigner_rows": ([P, P, I32, I32, I32, P], I32),
    "singa_gather_rotate_fwd": ([P, P, P, P, P, P, P, I32, I32, I32, I32, P], I32),
    "singa_gather_rotate_bwd": ([P] * 13 + [I32] * 6 + [P], I32),
    "singa_rotate_back_scatter_fwd": ([C.POINTER(Seg), I32, P, P, P, P, I32, I32, I32, I32, I32, I32, F32, P], I32),
    "singa_rotate_back_scatter_bwd": ([P, C.POINTER(Seg), C.POINTER(Seg), I32, P, P, P, P, I32, I32, I32, I32, I32,
                                       I32, F32, P], I32),
    "singa_alpha_logits_nslots": ([I32], I32),
    "singa_alpha_logits_fwd": ([P, C.c_longlong, P, P, P, P, I32, I32, I32, F32, P], I32),
    "singa_alpha_logits_bwd": ([P, C.c_longlong, P, P, P, P, P, P, I32, I32, I32, F32, P], I32),
    "singa_alpha_logits_bwd_ld": ([P, C.c_longlong, P, P, P, P, P, C.c_longlong, P, I32, I32, I32, F32, P], I32),
    "singa_segment_softmax_fwd": ([P, P, P, I32, I32, F32, I32, P], I32),
    "singa_segment_softmax_bwd": ([P, P, P, P, I32, I32, I32, P], I32),
    "singa_segment_wsum_fwd": ([P, P, P, P, I32, I32, I32, P], I32),
    "singa_segment_wsum_bwd": ([P, P, P, P, P, P, I32, I32, I32, P], I32),
    "singa_s2act_fwd": ([C.POINTER(Seg), I32, P, I64, P, P, P, I32, I32, I32, I32, P], I32),
    "singa_s2act_bwd": ([C.POINTER(Seg), I32, P, I64, P, P, P, P, P, I32, I32, I32, I32, P], I32),
    "singa_s2act_sep_fwd": ([C.POINTER(Seg), I32, P, I64, P, P, P, P, I32, I32, I32, P], I32),
    "singa_s2act_sep_bwd": ([C.POINTER(Seg), I32, P, I64, P, P, P, P, P, P, I32, I32, I32, P], I32),
    "singa_s2act_ffn_bwd": ([P, P, I64, P, P, P, P, P, P, I32, I32, I32, P], I32),
    "singa_s2act_sep_bwd_seg": ([C.POINTER(Seg), I32, P, I64, P, P, P, P, C.POINTER(Seg), P, I64, I32, I32, I32, P], I32),
    "singa_so3_rmsnorm_nparts": ([I32], I32),
    "singa_so3_rmsnorm_fwd": ([P, P, P, P, I32, I32, I32, F32, P], I32),
    "singa_so3_rmsnorm_bwd": ([P, P, P, P, P, P, I32, I32, I32, F32, P], I32),
    "singa_so3_rmsnorm_bwd_add": ([P, P, P, P, P, P, P, I32, I32, I32, F32, P], I32),
    "singa_edge_logits_fwd": ([P] * 7 + [I32, I32, I32, F32, P], I32),
    "singa_edge_logits_bwd": ([P] * 13 + [I32, I32, I32, F32, P], I32),
    "singa_gather_wsum_fwd": ([P] * 6 + [I32, I32, I32, P], I32),
    "singa_gather_wsum_bwd": ([P] * 12 + [I32, I32, I32, P], I32),
    "singa_bias_ssp_fwd": ([P] * 3 + [I64, I32, P], I32),
    "singa_bias_ssp_bwd": ([P] * 4 + [I64, I32, P], I32),
    "singa_dec_self_attn": ([P] * 10 + [I32, I32, P, F32, P], I32),
    "singa_dec_cross_attn": ([P] * 10 + [I32, I32, I32, P, F32, P], I32),
    "singa_dec_ffn": ([P] * 7 + [I32, P, F32, P], I32),
    "singa_edge_mlp_fwd": ([P] * 11 + [I32, I32, I32, I32, P], I32),
    "singa_edge_mlp_bwd_nparts": ([I32, I32], I32),
    "singa_edge_mlp_bwd": ([P] * 6 + [I32, I32, I32, P], I32),
    "singa_masked_softmax_fwd": ([P, P, I64, I64, P, I32, I32, I32, I32, F32, P], I32),
    "singa_masked_softmax_bwd": ([P, P, P, I64, I64, P, I32, I32, I32, I32, F32, P], I32),
    "singa_attn_fwd": ([P, P, P, P, I64, I64, P, P, I32, I32, I32, I32, I32, I32, I32, I64, I64, I64, F32, P], I32),
    "singa_attn_bwd": ([P, P, P, P, I64, I64] + [P] * 7 + [I32, I32, I32, I32, I32, I32, I32, I64, I64, I64, F32, P], I32),
    "singa_ln256_nparts": ([I64], I32),
    "singa_ln256_fwd": ([P] * 5 + [I64, I32, F32, P], I32),
    "singa_ln256_bwd": ([P] * 6 + [I64, I32, F32, P], I32),
    "singa_ln_silu_nparts": ([I64], I32),
    "singa_ln_silu_fwd": ([P] * 4 + [I64, I32, F32, P], I32),
    "singa_ln_silu_bwd": ([P] * 6 + [I64, I32, F32, P], I32),
    "singa_colsum_work": ([C.c_longlong, I32], C.c_longlong),
    "singa_colsum": ([P, C.c_longlong, C.c_longlong, I32, P, P, P], I32),
    "singa_so3_skinny_nparts": ([I32, I32, I32], I32),
    "singa_so3_skinny_expand": ([P, P, I64, I64, I64, P, P, I32, I32, I32, P], I32),
    "singa_so3_skinny_reduce": ([P, P, P, I32, I32, I32, I32, I32, P], I32),
    "singa_colsum_multi_work": ([C.c_longlong, I32], C.c_longlong),
    "singa_colsum_multi": ([I32, P, P, P, P, P, I32, P, P, P, C.c_longlong, P], I32),
    "singa_block_weight_fwd": ([P, P, I32, I32, P], I32),
    "singa_block_weight_bwd": ([P, P, I32, I32, I32, P], I32),
    "singa_rowdot_nparts": ([C.c_longlong], I32),
    "singa_rowdot_fwd": ([P, P, P, C.c_longlong, I32, F32, P], I32),
    "singa_rowdot_bwd": ([P, P, P, P, P, C.c_longlong, I32, F32, P], I32),
    "singa_lap_pe_work": ([I32, I32], I32),
    "singa_lap_pe": ([P] * 8 + [I32, I32, I32, P], I32),
    "singa_adam_step": ([P, P, P, P, P, P, P, I32, I32, P, P, F32, F32, F32, P], I32),
    "singa_grad_norm": ([P, P, P, P, I32, I32, P, P, P], I32),
    "singa_gemm_f32": ([C.POINTER(Gemm), I32, I32, I32, I32, P], I32),
    "singa_cgemm3m_f32": ([C.POINTER(CGemm), I32, I32, I32, I32, P], I32),
    "singa_knn_graph": ([P, P, P, I32, I32, I32, I32, P, P, P], I32),
    "singa_knn_edge_attr": ([P, P, C.c_longlong, P, C.c_float, P, I32, I32, P], I32),
    "singa_prof_enable": ([I32], I32),
    "singa_prof_hint_edges": ([I32], I32),
    "singa_prof_collect": ([P, P, P, I32], I32),
    "singa_prof_collect_tagged": ([P, P, P, P, I32], I32),
    "singa_prof_stamps": ([P, I32], I32),
    "singa_prof_read_stamps": ([P, P, P, P, P, I32], I32),
    "singa_prof_reset": ([], I32),
    "singa_calib_copy": ([P, P, C.c_longlong, P], I32),
    "singa_calib_copy16": ([P, P, C.c_longlong, I32, I32, P], I32),
}

EXPORTS = tuple(_SIGS)

# include/singa_hip_lab.h: test / lab switches, bound as well (tests/, tools/lab/ call them) but not part of the drop-in ABI
_LAB_SIGS = {
    "singa_so3_skinny_variant": ([I32], I32),
    "singa_gemm_occupancy": ([I32, I32, I32], I32),
    "singa_gemm_force_cfg": ([I32], I32),
    "singa_lap_pe_fsi_min": ([I32], I32),
}
LAB_EXPORTS = tuple(_LAB_SIGS)


def bind(path):
    import torch  # noqa: F401  - the HIP runtime bundled with PyTorch must be the one this library resolves against
    lib = C.CDLL(path)
    for name, (args, res) in list(_SIGS.items()) + list(_LAB_SIGS.items()):
        fn = getattr(lib, name)          # AttributeError if the library does not export a declared symbol
        fn.argtypes = args
        fn.restype = res
    return lib


def segs(items):
    """items: list of (ptr:int, ld:int, rows:int) -> (ctypes array, n)."""
    arr = SegArr()
    for i, (p, ld, rows) in enumerate(items):
        arr[i].ptr, arr[i].ld, arr[i].rows = p, ld, rows
    return arr, len(items)


def check(lib, code, what):
    if code != 0:
        raise RuntimeError(f"{what} failed ({code}): {lib.singa_last_error_string().decode()}")
