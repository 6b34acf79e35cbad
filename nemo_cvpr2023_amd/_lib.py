"""ctypes binding of ``libnemo_hip.so`` (the C ABI declared in ``include/nemo_hip.h``).

The library is built in-tree by ``__graft_entry__.build()`` / ``csrc/Makefile``.  There is NO
fallback: if the shared object is missing or a kernel call fails, this module raises.
"""
from __future__ import annotations

import ctypes
import os
from ctypes import POINTER, Structure, c_double, c_float, c_int32, c_int64, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('NEMO_HIP_LIB') or os.path.join(_HERE, 'libnemo_hip.so')    # env: kernel-variant A/B runs

ADAM_MAX_SEG = 16


class ColsumDesc(Structure):
    _fields_ = [('X', c_void_p), ('M', c_int64), ('N', c_int64), ('ldx', c_int64), ('out', c_void_p)]


class GemmProblem(Structure):
    _fields_ = [('transA', c_int32), ('transB', c_int32), ('M', c_int64), ('N', c_int64), ('K', c_int64),
                ('A', c_void_p), ('lda', c_int64), ('B', c_void_p), ('ldb', c_int64), ('C', c_void_p), ('ldc', c_int64),
                ('alpha', c_float), ('out_mode', c_int32)]


class CastXpDesc(Structure):
    _fields_ = [('src', c_void_p), ('rows', c_int64), ('cols', c_int64), ('lds', c_int64), ('dst', c_void_p), ('ldd', c_int64),
                ('dstT', c_void_p), ('lddT', c_int64), ('scale', c_float), ('meta', c_void_p)]


class GemmXpProblem(Structure):
    _fields_ = [('M', c_int64), ('N', c_int64), ('K', c_int64), ('A', c_void_p), ('lda', c_int64), ('B', c_void_p), ('ldb', c_int64),
                ('C', c_void_p), ('ldc', c_int64), ('alpha', c_float), ('out_mode', c_int32), ('metaA', c_void_p), ('metaB', c_void_p)]


class AbsmaxDesc(Structure):
    _fields_ = [('src', c_void_p), ('rows', c_int64), ('cols', c_int64), ('lds', c_int64), ('meta', c_void_p), ('overwrite', c_int32)]


class AdamSeg(Structure):
    _fields_ = [('offset', c_int64), ('numel', c_int64), ('lr', c_float), ('weight_decay', c_float),
                ('step_size', c_float), ('bias_corr2_sqrt', c_float), ('adamw', c_int32), ('step', c_int32)]


i32, i64, f32, ptr = c_int32, c_int64, c_float, c_void_p

ABI_VERSION = 18        # NEMO_ABI_VERSION of include/nemo_hip.h this binding was written against

# name -> (restype, argtypes).  Mirrors include/nemo_hip.h one to one (tests check the symbol list).
SIGNATURES = {
    'nemo_abi_version': (i32, []),
    'nemo_reduce_ws_bytes': (i64, [i64, i64]),
    'nemo_reduce_ws_bind': (i32, [ptr, i64]),
    'nemo_reduce_scratch_reset': (i32, []),
    'nemo_reduce_fallbacks': (i64, []),
    'nemo_gemm_f32': (i32, [i32, i32, i64, i64, i64, ptr, i64, ptr, i64, ptr, i64, ptr, i32, ptr, i64, i32,
                            f32, i32, i32, ptr, i64, ptr]),
    'nemo_gemm_f32_colsum': (i32, [i32, i32, i64, i64, i64, ptr, i64, ptr, i64, ptr, i64, ptr, i32, ptr, i64, i32, f32,
                                   ptr, i64, ptr, i64, ptr]),
    'nemo_gemm_bf16': (i32, [i32, i32, i64, i64, i64, ptr, i64, ptr, i64, ptr, i64, ptr, i32, ptr, i64, i32,
                             f32, i32, i32, ptr, i64, ptr]),
    'nemo_gemm_bf16mem': (i32, [i64, i64, i64, ptr, i64, ptr, i64, ptr, i64, ptr, i32, ptr, i64, i32, f32, i32, ptr, i64, ptr,
                                i64, ptr, i64, ptr, i64, ptr]),
    'nemo_gemm_colsum_rows': (i64, [i64]),
    'nemo_gemm_f32_b16out': (i32, [i32, i32, i64, i64, i64, ptr, i64, ptr, i64, ptr, i64, ptr, i32, ptr, i64, ptr, i64, ptr, i64,
                                   ptr]),
    'nemo_cast_bf16': (i32, [i64, i64, ptr, i64, ptr, i64, i32, ptr]),
    'nemo_cast_bf16_split3': (i32, [i64, i64, ptr, i64, ptr, i64, i32, ptr]),
    'nemo_xp_ld': (i64, [i32, i64]),
    'nemo_gemm_xp': (i32, [i32, i64, i64, i64, ptr, i64, ptr, i64, ptr, i64, ptr, i32, ptr, i64, i32, f32, i32, ptr, i64, ptr, i64,
                           f32, ptr, i64, ptr, ptr, ptr, ptr, ptr, ptr, i64, ptr]),
    'nemo_cast_xp': (i32, [i32, i32, POINTER(CastXpDesc), ptr]),
    'nemo_gemm_xp_grouped': (i32, [i32, i32, POINTER(GemmXpProblem), ptr, i64, ptr]),
    'nemo_absmax_multi': (i32, [i32, POINTER(AbsmaxDesc), ptr]),
    'nemo_gemm_grouped_f32': (i32, [i32, POINTER(GemmProblem), ptr, i64, ptr]),
    'nemo_gemm_grouped_bf16': (i32, [i32, POINTER(GemmProblem), ptr, i64, ptr]),
    'nemo_colsum_f32': (i32, [ptr, i64, i64, i64, ptr, ptr]),
    'nemo_colsum_multi': (i32, [i32, POINTER(ColsumDesc), ptr]),
    'nemo_phase_embed_fwd': (i32, [i64, i64, i64, i64, i64, i64, ptr, ptr, ptr, ptr, ptr, i64, ptr, ptr, ptr,
                                   i32, ptr, i64, ptr, ptr, ptr, ptr]),
    'nemo_phase_embed_bwd': (i32, [i64, i64, i64, i64, i64, i64, ptr, ptr, ptr, ptr, ptr, i64, ptr, i32, ptr,
                                   ptr, i64, ptr, ptr, ptr, ptr, ptr, i32, ptr]),
    'nemo_phase_embed_bwd_colsum': (i32, [i64, i64, i64, i64, i64, i64, ptr, ptr, ptr, ptr, ptr, i64, ptr, i32, ptr,
                                          ptr, i64, ptr, ptr, ptr, ptr, ptr, i32, POINTER(ColsumDesc), i32, ptr]),
    'nemo_rot6d_fwd': (i32, [i64, i64, ptr, i64, i32, ptr, ptr, ptr]),
    'nemo_rot6d_bwd': (i32, [i64, i64, ptr, i64, i32, ptr, ptr, ptr, i64, ptr]),
    'nemo_pose_bwd_fused': (i32, [i64, ptr, i64, i32, ptr, ptr, ptr, i64, ptr, ptr, f32, ptr, i64, i32, ptr, ptr]),
    'nemo_rotmat_to_aa': (i32, [i64, ptr, i32, ptr, ptr]),
    'nemo_rodrigues_fwd': (i32, [i64, ptr, i32, ptr, ptr]),
    'nemo_rodrigues_bwd': (i32, [i64, ptr, ptr, ptr, ptr]),
    'nemo_ctx_create': (i32, [POINTER(c_void_p), i64, ptr, ptr, ptr, ptr, ptr, ptr, i64, ptr, i64, ptr, i64,
                              ptr]),
    'nemo_ctx_set_betas': (i32, [ptr, ptr]),
    'nemo_ctx_destroy': (i32, [ptr]),
    'nemo_ctx_num_verts': (i64, [ptr]),
    'nemo_ctx_skin_nnz': (i32, [ptr]),
    'nemo_ctx_split_ok': (i32, [ptr]),
    'nemo_ctx_skin_mfma_ok': (i32, [ptr]),
    'nemo_ctx_vp_bound': (f32, [ptr]),
    'nemo_ctx_skin_sparse': (i32, [ptr]),
    'nemo_ctx_set_skin_sparse': (i32, [ptr, i32]),
    'nemo_ctx_nq': (i64, [ptr]),
    'nemo_ctx_C1': (ptr, [ptr]),
    'nemo_ctx_c0': (ptr, [ptr]),
    'nemo_ctx_posedirs': (ptr, [ptr]),
    'nemo_ctx_posedirs_ld': (i64, [ptr]),
    'nemo_ctx_v_shaped': (ptr, [ptr]),
    'nemo_fk_fwd': (i32, [ptr, i64, ptr, ptr, ptr, ptr, i64, ptr]),
    'nemo_fk_bwd': (i32, [ptr, i64, ptr, ptr, ptr, ptr, ptr, i64, ptr, ptr]),
    'nemo_kp_fwd': (i32, [ptr, i64, i64, i64, ptr, ptr, ptr, i64, ptr, i64, i32, ptr, ptr, ptr, ptr, ptr,
                          f32, f32, f32, i32, i32, ptr, ptr, ptr, ptr, ptr, ptr]),
    'nemo_kp_finalize': (i32, [i64, i64, i32, i32, ptr, ptr, ptr, ptr]),
    'nemo_kp_bwd': (i32, [ptr, i64, i64, i64, ptr, ptr, ptr, i64, ptr, i64, i32, ptr, ptr, ptr, ptr, ptr,
                          f32, f32, f32, i32, i32, ptr, ptr, f32, ptr, ptr, ptr, ptr, i64, ptr, ptr]),
    'nemo_kp_bwd_ex': (i32, [ptr, i64, i64, i64, ptr, ptr, ptr, i64, ptr, i64, i32, ptr, ptr, ptr, ptr, ptr,
                             f32, f32, f32, i32, i32, ptr, ptr, f32, ptr, ptr, ptr, ptr, i64, ptr, ptr, ptr, ptr]),
    'nemo_kp_fwd_bwd': (i32, [ptr, i64, i64, i64, ptr, ptr, ptr, i64, ptr, i64, i32, ptr, ptr, ptr, ptr, ptr,
                              f32, f32, f32, i32, i32, ptr, f32, ptr, ptr, ptr, ptr, ptr, ptr, ptr, ptr, i64, ptr, ptr, ptr]),
    'nemo_smooth_fwd_bwd': (i32, [i64, i64, i64, ptr, f32, ptr, ptr, ptr]),
    'nemo_project': (i32, [i64, i64, i64, ptr, ptr, ptr, f32, f32, f32, ptr, ptr]),
    'nemo_skin_vertices': (i32, [ptr, i64, ptr, i64, ptr, ptr, i64, ptr, ptr]),
    'nemo_v2v_skin_l1': (i32, [ptr, i64, ptr, i64, ptr, ptr, ptr, i64, ptr, ptr]),
    'nemo_v2v_fused_ws_bytes': (i64, [ptr, i64]),
    'nemo_v2v_fused': (i32, [ptr, i64, ptr, i64, ptr, ptr, ptr, i64, ptr, ptr, i64, ptr]),
    'nemo_v2v_fused_split': (i32, [ptr, i64, ptr, i64, ptr, ptr, ptr, i64, ptr, ptr, i64, ptr]),
    'nemo_v2v_fused_splitmem': (i32, [ptr, i64, ptr, i64, ptr, ptr, ptr, i64, i64, ptr, ptr, i64, ptr]),
    'nemo_v2v_fused_splitxp': (i32, [ptr, i64, ptr, i64, ptr, ptr, ptr, i64, ptr, ptr, i64, ptr]),
    'nemo_gemm_f16x2mem_adj': (i32, [i64, i64, i64, ptr, i64, i64, ptr, i64, i64, ptr, i64, f32, i32, ptr, i64, ptr]),
    'nemo_v2v_fused_bf16': (i32, [ptr, i64, ptr, i64, ptr, ptr, ptr, i64, ptr, ptr, i64, ptr]),
    'nemo_v2v_combine': (i32, [ptr, i64, ptr, ptr, i64, ptr]),
    'nemo_v2v_fused_bf16mem': (i32, [ptr, i64, ptr, i64, ptr, ptr, ptr, i64, ptr, ptr, i64, ptr]),
    'nemo_v2v_prep_fwd': (i32, [i64, ptr, ptr, ptr, ptr, ptr, ptr]),
    'nemo_v2v_prep_fwd_dec': (i32, [i64, ptr, ptr, ptr, i64, ptr, ptr, ptr, ptr]),
    'nemo_v2v_prep_bwd': (i32, [i64, ptr, ptr, f32, ptr, ptr, ptr]),
    'nemo_kl_fwd_bwd': (i32, [i64, i64, ptr, i64, ptr, ptr, i64, ptr, ptr]),
    'nemo_publish_scalars': (i32, [ptr, i32, ptr, ptr, ptr]),
    'nemo_gmm_fwd_bwd': (i32, [i64, i64, i64, ptr, i64, ptr, ptr, ptr, ptr, ptr, ptr, f32, ptr, i64, ptr, ptr]),
    'nemo_pose3d_fwd_bwd': (i32, [i64, i64, ptr, i64, ptr, ptr, ptr, ptr, i64, ptr, f32, ptr, i64, ptr, ptr]),
    'nemo_adam_step': (i32, [i32, POINTER(AdamSeg), ptr, ptr, ptr, ptr, f32, f32, f32, ptr]),
    'nemo_adam_step_dev': (i32, [i32, ptr, i64, ptr, ptr, ptr, ptr, f32, f32, f32, ptr]),
    'nemo_adam_step_dev_if': (i32, [i32, ptr, i64, ptr, ptr, ptr, ptr, f32, f32, f32, ptr, ptr]),
    'nemo_nan_count': (i32, [ptr, i64, ptr, ptr]),
    'nemo_seq_gather': (i32, [ptr, ptr, i64, ptr, ptr, ptr, ptr]),
    'nemo_seq_log': (i32, [ptr, i32, ptr, i64, ptr, ptr]),
    'nemo_sqmean_fwd_bwd': (i32, [i64, ptr, ptr, ptr, f32, ptr]),
    'nemo_step_begin': (i32, [ptr, i64, ptr, i64, ptr, i32, c_double, c_double, ptr]),
    'nemo_phase_embed_fwd_begin': (i32, [i64, i64, i64, i64, i64, i64, ptr, ptr, ptr, ptr, ptr, i64, ptr, ptr, ptr,
                                         i32, ptr, i64, ptr, ptr, ptr, ptr, i64, ptr, i64, ptr, i32, c_double, c_double, i32, ptr, ptr]),
    'nemo_scale_neg_rowsum': (i32, [i64, i64, ptr, i64, ptr, ptr]),
}

_lib = None


class NemoHipError(RuntimeError):
    pass


def load():
    """dlopen the library and attach prototypes.  Raises if the HIP extension is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise NemoHipError(
            f'{LIB_PATH} not found: the HIP extension is not built. Run `python -c "import '
            f'__graft_entry__ as g; g.build()"` (or `make -C nemo_cvpr2023_amd/csrc`). '
            'There is no CPU fallback by design.')
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    ver = lib.nemo_abi_version()
    if ver != ABI_VERSION:
        raise NemoHipError(f'libnemo_hip.so ABI version {ver} != {ABI_VERSION} (stale build?)')
    _lib = lib
    return lib


def load_for_engine():
    """The library as FitEngine sees it (a seam for measurement tools: tools/ablate.py wraps it)."""
    return load()


def check(rc: int, what: str):
    if rc != 0:
        kind = 'invalid argument' if rc < 0 else f'hipError_t {rc}'
        raise NemoHipError(f'{what} failed: {kind}')


def dptr(t):
    """Device pointer of a tensor (or None)."""
    if t is None:
        return None
    return t.data_ptr()
