"""Tensor-in / tensor-out wrappers over the C ABI, used only by the GPU tests."""
import ctypes

import torch

from nemo_cvpr2023_amd import _lib
from nemo_cvpr2023_amd._lib import check, dptr

DEV = 'cuda:0'


_RED_WS = []


def st():
    """The stream argument of a test launch; also (re)binds the tests' reduction arena (include/nemo_hip.h nemo_reduce_ws_bind) so
    that every direct kernel call takes the ordered, deterministic path from the arena's start."""
    L = _lib.load()
    if not _RED_WS:
        _RED_WS.append(torch.zeros(int(L.nemo_reduce_ws_bytes(262144, 256)) // 4, device=DEV))
    check(L.nemo_reduce_ws_bind(dptr(_RED_WS[0]), _RED_WS[0].numel() * 4), 'nemo_reduce_ws_bind')
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def dev(t, dtype=torch.float32):
    return torch.as_tensor(t).to(DEV, dtype).contiguous()


_GEMM_WS = {}


def gemm_ws(nbytes=64 << 20):
    if nbytes not in _GEMM_WS:
        _GEMM_WS[nbytes] = torch.zeros(nbytes // 4, device=DEV)
    return _GEMM_WS[nbytes]


def gemm(A, B, ta=0, tb=0, bias=None, act=0, mask=None, mask_mode=0, alpha=1.0, out_mode=0, split_k=1,
         C=None, ws=True):
    L = _lib.load()
    M = A.shape[1] if ta else A.shape[0]
    K = A.shape[0] if ta else A.shape[1]
    N = B.shape[0] if tb else B.shape[1]
    if C is None:
        C = torch.zeros(M, N, device=DEV)
    check(L.nemo_gemm_f32(ta, tb, M, N, K, dptr(A), A.stride(0), dptr(B), B.stride(0), dptr(C), C.stride(0),
                          dptr(bias), act, dptr(mask), mask.stride(0) if mask is not None else 0, mask_mode,
                          alpha, out_mode, split_k, dptr(gemm_ws()) if ws else None,
                          gemm_ws().numel() * 4 if ws else 0, st()), 'gemm')
    return C


def xp_ld(fmt, k):
    return int(_lib.load().nemo_xp_ld(fmt, k))


def absmax_meta(*srcs):
    """One scale record (include/nemo_hip.h: [0] scale, [2, 34) absmax slots) per fp32 matrix: a (n, 64) float tensor holding the
    absmax (nemo_absmax_multi)."""
    L = _lib.load()
    meta = torch.zeros(len(srcs), 64, device=DEV)
    d = (_lib.AbsmaxDesc * len(srcs))()
    for i, x in enumerate(srcs):
        d[i].src, d[i].rows, d[i].cols, d[i].lds, d[i].meta = dptr(x), x.shape[0], x.shape[1], x.stride(0), meta[i].data_ptr()
    check(L.nemo_absmax_multi(len(srcs), d, st()), 'absmax_multi')
    return meta


def meta_scale(meta):
    return float(meta[0])


def meta_amax(meta):
    return float(meta[2:34].max())


def cast_xp(fmt, src, plain=True, transposed=False, scale=1.0, meta=None):
    """fp32 (rows x cols) -> its xp copies (int16 tensors): (plain or None, transposed or None).  fmt 2 with `meta` (a float[2]
    device record whose [0] holds the absmax): the scale is chosen on the device and left in meta[1]."""
    L = _lib.load()
    rows, cols = src.shape
    d = (_lib.CastXpDesc * 1)()
    dst = torch.zeros(rows, xp_ld(fmt, cols), dtype=torch.int16, device=DEV) if plain else None
    dstT = torch.zeros(cols, xp_ld(fmt, rows), dtype=torch.int16, device=DEV) if transposed else None
    d[0].src, d[0].rows, d[0].cols, d[0].lds = dptr(src), rows, cols, src.stride(0)
    d[0].dst, d[0].ldd = dptr(dst), dst.stride(0) if plain else 0
    d[0].dstT, d[0].lddT = dptr(dstT), dstT.stride(0) if transposed else 0
    d[0].scale = scale
    d[0].meta = dptr(meta)
    check(L.nemo_cast_xp(fmt, 1, d, st()), 'cast_xp')
    return dst, dstT


def xp_decode(fmt, x, k, scale=1.0):
    """The fp64 value an xp matrix (rows x ld int16) holds: sum of its pieces / scale; (rows x k)."""
    rows = x.shape[0]
    kb = (k + 31) // 32
    v = x[:, :kb * 32 * fmt].reshape(rows, kb, fmt, 32)
    v = v.view(torch.bfloat16 if fmt == 3 else torch.float16).double().sum(2).reshape(rows, kb * 32)
    return v[:, :k] / scale


def gemm_xp(fmt, Ax, Bx, M, N, K, C=None, bias=None, act=0, maskx=None, alpha=1.0, out_mode=0, want_cx=False, want_cxt=False,
            out_scale=1.0, colsum=False, ws=True, metaA=None, metaB=None, metaBias=None, metaOut=None, mask_mode=1):
    L = _lib.load()
    if C is None and not (want_cx or want_cxt):
        C = torch.zeros(M, N, device=DEV)
    Cx = torch.zeros(M, xp_ld(fmt, N), dtype=torch.int16, device=DEV) if want_cx else None
    CxT = torch.zeros(N, xp_ld(fmt, M), dtype=torch.int16, device=DEV) if want_cxt else None
    cs = torch.zeros(int(L.nemo_gemm_colsum_rows(M)), N, device=DEV) if colsum else None
    check(L.nemo_gemm_xp(fmt, M, N, K, dptr(Ax), Ax.stride(0), dptr(Bx), Bx.stride(0), dptr(C), C.stride(0) if C is not None else 0,
                         dptr(bias), act, dptr(maskx), maskx.stride(0) if maskx is not None else 0, mask_mode if maskx is not None else 0,
                         alpha, out_mode, dptr(Cx), Cx.stride(0) if want_cx else 0, dptr(CxT), CxT.stride(0) if want_cxt else 0,
                         out_scale, dptr(cs), cs.stride(0) if colsum else 0, dptr(metaA), dptr(metaB), dptr(metaBias), dptr(metaOut), None,
                         dptr(gemm_ws()) if ws else None, gemm_ws().numel() * 4 if ws else 0, st()), 'gemm_xp')
    return C, Cx, CxT, cs
