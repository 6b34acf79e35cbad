"""Tensor-in / tensor-out wrappers over the C ABI, used only by the GPU tests."""
import ctypes

import torch

from nemo_cvpr2023_amd import _lib
from nemo_cvpr2023_amd._lib import check, dptr

DEV = 'cuda:0'


def st():
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
