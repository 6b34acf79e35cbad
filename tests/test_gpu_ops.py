"""GPU parity tests: every HIP kernel (through the C ABI) against the CPU oracle on seeded inputs.
Tolerances are relative to the tensor's max magnitude (fp32; north-star gate is 1e-4)."""
import ctypes
import math

import numpy as np
import pytest
import torch

from conftest import load_golden, rel_err
from nemo_cvpr2023_amd import synthetic as syn

pytestmark = pytest.mark.gpu

TOL = 2e-5


@pytest.fixture(scope='module')
def L():
    from nemo_cvpr2023_amd import _lib
    assert torch.cuda.is_available(), 'GPU tests need an MI355X'
    return _lib.load()


def _ops():
    import hipops
    return hipops


# ------------------------------------------------------------------------------------------ GEMM
@pytest.mark.parametrize('ta,tb', [(0, 0), (0, 1), (1, 0), (1, 1)])
@pytest.mark.parametrize('M,N,K', [(70, 33, 45), (257, 130, 207), (300, 1000, 1000), (1, 3, 7), (129, 64, 16)])
def test_gemm_layouts(L, ta, tb, M, N, K):
    H = _ops()
    g = torch.Generator().manual_seed(M * 7 + N * 3 + K + ta * 2 + tb)
    A = torch.randn((K, M) if ta else (M, K), generator=g)
    B = torch.randn((N, K) if tb else (K, N), generator=g)
    ref = (A.T if ta else A).double() @ (B.T if tb else B).double()
    C = H.gemm(H.dev(A), H.dev(B), ta, tb)
    assert rel_err(C, ref) < TOL


def test_gemm_big_tile_path(L):
    H = _ops()
    g = torch.Generator().manual_seed(5)
    A, B = torch.randn(2600, 207, generator=g), torch.randn(207, 3000, generator=g)
    bias = torch.randn(3000, generator=g)
    C = H.gemm(H.dev(A), H.dev(B), bias=H.dev(bias), split_k=0)       # library-chosen plan
    assert rel_err(C, A.double() @ B.double() + bias.double()) < TOL


def test_gemm_epilogues(L):
    H = _ops()
    g = torch.Generator().manual_seed(9)
    A, B = torch.randn(150, 90, generator=g), torch.randn(70, 90, generator=g)
    bias, mask = torch.randn(70, generator=g), torch.randn(150, 70, generator=g)
    pre = A.double() @ B.double().T + bias.double()
    dA, dB, db, dm = H.dev(A), H.dev(B), H.dev(bias), H.dev(mask)
    assert rel_err(H.gemm(dA, dB, 0, 1, bias=db, act=1), torch.relu(pre)) < TOL
    assert rel_err(H.gemm(dA, dB, 0, 1, bias=db, act=2), torch.nn.functional.leaky_relu(pre, 0.01)) < TOL
    assert rel_err(H.gemm(dA, dB, 0, 1, mask=dm, mask_mode=1), (A.double() @ B.double().T) * (mask > 0)) < TOL
    lm = torch.where(mask > 0, torch.ones_like(mask), torch.full_like(mask, 0.01)).double()
    assert rel_err(H.gemm(dA, dB, 0, 1, mask=dm, mask_mode=2), (A.double() @ B.double().T) * lm) < TOL
    C0 = torch.randn(150, 70, generator=g)
    C = H.dev(C0).clone()
    H.gemm(dA, dB, 0, 1, alpha=0.5, out_mode=1, C=C)
    assert rel_err(C, C0.double() + 0.5 * (A.double() @ B.double().T)) < TOL
    C = H.dev(C0).clone()
    H.gemm(dA, dB, 0, 1, out_mode=2, split_k=4, bias=db, C=C)
    assert rel_err(C, C0.double() + pre) < TOL
    # strided views (ld > cols), as the engine uses for AA[:, 3:66]
    big = H.dev(torch.randn(150, 200, generator=g))
    sub = big[:, 5:95]
    assert rel_err(H.gemm(sub, dB, 0, 1), sub.cpu().double() @ B.double().T) < TOL


@pytest.mark.parametrize('ta,tb', [(0, 0), (0, 1), (1, 0), (1, 1)])
@pytest.mark.parametrize('M,N,K', [(2401, 1000, 1000), (130, 72, 100), (64, 20, 207), (4100, 3000, 208)])
def test_gemm_vector_path(L, ta, tb, M, N, K):
    """16-byte aligned operands take the dwordx4 staging path; ragged M/N/K tails included."""
    H = _ops()
    g = torch.Generator().manual_seed(M + N + K + 2 * ta + tb)
    lda = ((M if ta else K) + 3) // 4 * 4
    ldb = ((K if tb else N) + 3) // 4 * 4
    Af = torch.randn((K if ta else M), lda, generator=g)
    Bf = torch.randn((N if tb else K), ldb, generator=g)
    A, B = Af[:, :(M if ta else K)], Bf[:, :(K if tb else N)]
    ref = (A.T if ta else A).double() @ (B.T if tb else B).double()
    dA, dB = H.dev(Af)[:, :A.shape[1]], H.dev(Bf)[:, :B.shape[1]]
    assert dA.data_ptr() % 16 == 0 and dA.stride(0) % 4 == 0
    C = H.gemm(dA, dB, ta, tb)
    assert rel_err(C, ref) < TOL


@pytest.mark.parametrize('tile', [64, 128])
@pytest.mark.parametrize('ta,tb', [(0, 0), (0, 1), (1, 0), (1, 1)])
@pytest.mark.parametrize('M,N,K,split', [(300, 1000, 1000, 3), (207, 2400, 2070, 8), (1000, 148, 2401, 5),
                                         (70, 50, 64, 2)])
def test_gemm_split_k_in_launch_combine(L, monkeypatch, tile, ta, tb, M, N, K, split):
    """K slices combined by the last-arriving block (fixed order): fused bias + ReLU epilogue, C += mode,
    bit-identical results run to run, tickets returned to zero (a second launch works)."""
    H = _ops()
    monkeypatch.setenv('NEMO_GEMM_TILE', str(tile))
    g = torch.Generator().manual_seed(M + N + K + 2 * ta + tb)
    lda = ((M if ta else K) + 3) // 4 * 4
    ldb = ((K if tb else N) + 3) // 4 * 4
    Af = torch.randn((K if ta else M), lda, generator=g)
    Bf = torch.randn((N if tb else K), ldb, generator=g)
    A, B = Af[:, :(M if ta else K)], Bf[:, :(K if tb else N)]
    bias = torch.randn(N, generator=g)
    pre = (A.T if ta else A).double() @ (B.T if tb else B).double() + bias.double()
    dA, dB, db = H.dev(Af)[:, :A.shape[1]], H.dev(Bf)[:, :B.shape[1]], H.dev(bias)
    C1 = H.gemm(dA, dB, ta, tb, bias=db, act=1, split_k=split)
    assert rel_err(C1, torch.relu(pre)) < TOL
    C2 = H.gemm(dA, dB, ta, tb, bias=db, act=1, split_k=split)
    assert torch.equal(C1, C2)
    C0 = torch.randn(M, N, generator=g)
    C = H.dev(C0).clone()
    H.gemm(dA, dB, ta, tb, bias=db, out_mode=1, split_k=split, C=C)
    assert rel_err(C, C0.double() + pre) < TOL
    assert int(H.gemm_ws()[:4096].view(torch.int32).abs().sum()) == 0


@pytest.mark.parametrize('M,N,K', [(2400, 207, 20670), (1201, 207, 2100), (1100, 200, 4099), (8192, 207, 2048), (1003, 129, 2222),
                                   (300, 207, 20670), (512, 207, 20670), (257, 207, 5000)])
def test_blend_shape_adjoint_on_the_mixed_shape_tile(L, M, N, K):
    """csrc/gemm_adj.h (round 4): the TT product dPF (+)= dVP^T P^T from 256 samples on, on one 64 x 208 column tile per
    workgroup -- columns [0, 192) on v_mfma_f32_32x32x2_f32, the remainder on 16x16x4 -- with K slices combined in the
    launch: overwrite and accumulate modes against float64, NaN-poisoned row pads, bit-identical run to run, tickets
    returned to zero."""
    H = _ops()
    g = torch.Generator().manual_seed(M + N + K)
    lda, ldb = (M + 15) // 16 * 16 + 16, (K + 3) // 4 * 4 + 4
    Af, Bf = torch.randn(K, lda, generator=g), torch.randn(N, ldb, generator=g)
    Af[:, M:] = float('nan')
    Bf[:, K:] = float('nan')
    ref = Af[:, :M].T.double() @ Bf[:, :K].T.double()
    dA, dB = H.dev(Af)[:, :M], H.dev(Bf)[:, :K]
    C1 = H.gemm(dA, dB, 1, 1, split_k=0)
    assert rel_err(C1, ref) < TOL
    assert torch.equal(C1, H.gemm(dA, dB, 1, 1, split_k=0))
    C0 = torch.randn(M, 208, generator=g)
    C = H.dev(C0).clone()
    H.gemm(dA, dB, 1, 1, alpha=0.5, out_mode=1, split_k=0, C=C[:, :N])
    assert rel_err(C[:, :N], C0[:, :N].double() + 0.5 * ref) < TOL
    assert torch.equal(C[:, N:].cpu(), C0[:, N:])                  # columns beyond N untouched
    assert int(H.gemm_ws()[:4096].view(torch.int32).abs().sum()) == 0


@pytest.mark.parametrize('M,N,K', [(2400, 1000, 1000), (300, 1000, 1000), (1000, 1000, 300), (2400, 148, 1000),
                                   (207, 2400, 20670), (600, 512, 512), (300, 64, 512)])
def test_gemm_auto_plan(L, M, N, K):
    """split_k = 0: whatever (tile, split) the host cost model picks must give the same product."""
    H = _ops()
    g = torch.Generator().manual_seed(M + N + K)
    A, B = torch.randn(M, K, generator=g), torch.randn(N, K, generator=g)
    C = H.gemm(H.dev(A), H.dev(B), 0, 1, split_k=0)
    assert rel_err(C, A.double() @ B.double().T) < TOL


@pytest.mark.parametrize('ta,tb', [(0, 0), (0, 1), (1, 0), (1, 1)])
@pytest.mark.parametrize('M,N,K', [(2401, 1000, 1000), (1100, 1000, 512)])
def test_gemm_whole_tiles_plus_split_tail(L, ta, tb, M, N, K):
    """More than 256 tiles and not a multiple of 256: the plan keeps whole tiles for every full round of CUs and
    cuts only the remainder along K (GemmArgs::t0).  Fused epilogues, accumulation, determinism and the ticket
    counters are checked on that path."""
    H = _ops()
    g = torch.Generator().manual_seed(M + 3 * N + K + ta + 2 * tb)
    A = torch.randn((K, M) if ta else (M, K), generator=g)
    B = torch.randn((N, K) if tb else (K, N), generator=g)
    bias, mask = torch.randn(N, generator=g), torch.randn(M, N, generator=g)
    ref = (A.T if ta else A).double() @ (B.T if tb else B).double()
    dA, dB = H.dev(A), H.dev(B)
    C1 = H.gemm(dA, dB, ta, tb, split_k=0, bias=H.dev(bias), act=1)
    assert rel_err(C1, torch.relu(ref + bias.double())) < TOL
    C2 = H.gemm(dA, dB, ta, tb, split_k=0, bias=H.dev(bias), act=1)
    assert torch.equal(C1, C2)                                   # slices are summed in a fixed order
    assert rel_err(H.gemm(dA, dB, ta, tb, split_k=0, mask=H.dev(mask), mask_mode=1), ref * (mask > 0)) < TOL
    C0 = torch.randn(M, N, generator=g)
    C = H.dev(C0).clone()
    H.gemm(dA, dB, ta, tb, split_k=0, alpha=0.5, out_mode=1, C=C)
    assert rel_err(C, C0.double() + 0.5 * ref) < TOL
    assert int(H.gemm_ws()[:4096].view(torch.int32).abs().sum()) == 0


def test_gemm_rejects_bad_args(L):
    x = torch.zeros(4, 4, device='cuda')
    rc = L.nemo_gemm_f32(0, 0, 4, 4, 4, x.data_ptr(), 4, x.data_ptr(), 4, x.data_ptr(), 4, None, 1, None, 0, 0,
                         1.0, 2, 2, None, 0, None)
    assert rc < 0      # atomically accumulated slices with a non-linear epilogue
    rc = L.nemo_gemm_f32(0, 0, 4, 4, 4, x.data_ptr(), 4, x.data_ptr(), 4, x.data_ptr(), 4, None, 0, None, 0, 0,
                         1.0, 0, 2, None, 0, None)
    assert rc < 0      # in-launch combine without a workspace
    assert L.nemo_gemm_f32(0, 0, 0, 4, 4, None, 4, None, 4, x.data_ptr(), 4, None, 0, None, 0, 0, 1.0, 0, 1,
                           None, 0, None) == 0   # empty problem is a no-op


def test_colsum(L):
    H = _ops()
    X = torch.randn(1000, 77, generator=torch.Generator().manual_seed(1))
    out = torch.zeros(77, device='cuda')
    assert L.nemo_colsum_f32(H.dev(X).data_ptr(), 1000, 77, 77, out.data_ptr(), H.st()) == 0
    assert rel_err(out, X.double().sum(0)) < TOL


# ------------------------------------------------------------------------------------------ rotations
def test_rot6d_fwd_bwd_vs_oracle_and_golden(L):
    from oracle import ops
    H = _ops()
    g = load_golden('fn_rot6d_to_rotmat')
    x = torch.tensor(g['x'])              # (64, 6) incl. near-identity rows
    rows, J = 8, 8
    x144 = H.dev(x.reshape(rows, J * 6))
    R = torch.zeros(rows, J, 9, device='cuda')
    aa = torch.zeros(rows, J * 3, device='cuda')
    assert L.nemo_rot6d_fwd(rows, J, x144.data_ptr(), J * 6, 1, R.data_ptr(), aa.data_ptr(), H.st()) == 0
    assert rel_err(R.reshape(-1, 3, 3), g['out']) < TOL
    xo = x.clone().requires_grad_(True)
    Ro = ops.rot6d_to_rotmat(xo)
    aao = ops.rotmat_to_aa(Ro)
    assert rel_err(aa.reshape(-1, 3), aao.detach()) < 1e-4
    gen = torch.Generator().manual_seed(3)
    ctR, cta = torch.randn(64, 3, 3, generator=gen), torch.randn(64, 3, generator=gen)
    ((Ro * ctR).sum() + (aao * cta).sum()).backward()
    dx = torch.zeros(rows, J * 6, device='cuda')
    assert L.nemo_rot6d_bwd(rows, J, x144.data_ptr(), J * 6, 1, H.dev(ctR).data_ptr(), H.dev(cta).data_ptr(),
                            dx.data_ptr(), J * 6, H.st()) == 0
    assert rel_err(dx.reshape(-1, 6), xo.grad) < 1e-4
    # dR only (golden gradient from the real reference)
    assert L.nemo_rot6d_bwd(rows, J, x144.data_ptr(), J * 6, 1, H.dev(g['ct']).data_ptr(), None,
                            dx.data_ptr(), J * 6, H.st()) == 0
    assert rel_err(dx.reshape(-1, 6), g['grad_x']) < 1e-4


def test_rotmat_to_aa_all_branches(L):
    H = _ops()
    g = load_golden('fn_rotmat_to_aa')
    R = H.dev(g['R'])
    M = R.shape[0]
    aa = torch.zeros(M, 3, device='cuda')
    assert L.nemo_rotmat_to_aa(M, R.data_ptr(), 1, aa.data_ptr(), H.st()) == 0
    assert rel_err(aa, g['out']) < 1e-4
    eye = torch.eye(3, device='cuda').reshape(1, 9).contiguous()
    out = torch.ones(1, 3, device='cuda')
    assert L.nemo_rotmat_to_aa(1, eye.data_ptr(), 1, out.data_ptr(), H.st()) == 0
    assert np.array_equal(out.cpu().numpy(), g['out_identity'])
    g2 = load_golden('fn_matrot2aa')
    assert L.nemo_rotmat_to_aa(M, R.data_ptr(), 0, aa.data_ptr(), H.st()) == 0
    assert rel_err(aa, g2['out']) < 1e-4


def test_rotmat_to_aa_backward_all_branches(L):
    """Adjoint of the 4-branch quaternion path: feed R through an (invertible) rot6d so that the
    C entry point (rot6d_bwd with daa only) exercises it; compare with the reference gradient chain."""
    from oracle import ops
    H = _ops()
    g = load_golden('fn_rotmat_to_aa')
    R = torch.tensor(g['R'])
    x = torch.stack([R[:, 0, 0], R[:, 0, 1], R[:, 1, 0], R[:, 1, 1], R[:, 2, 0], R[:, 2, 1]], 1)
    M = x.shape[0]
    xo = x.clone().requires_grad_(True)
    aao = ops.rotmat_to_aa(ops.rot6d_to_rotmat(xo))
    ct = torch.tensor(g['ct'])
    (aao * ct).sum().backward()
    dx = torch.zeros(M, 6, device='cuda')
    assert L.nemo_rot6d_bwd(M, 1, H.dev(x).data_ptr(), 6, 1, None, H.dev(ct).data_ptr(), dx.data_ptr(), 6,
                            H.st()) == 0
    ref = xo.grad
    ok = torch.isfinite(ref).all(1)
    assert ok.sum() > 100
    assert rel_err(dx.cpu()[ok], ref[ok]) < 2e-4


def test_rodrigues(L):
    H = _ops()
    g = load_golden('fn_batch_rodrigues')
    th = H.dev(g['theta'])
    M = th.shape[0]
    R = torch.zeros(M, 9, device='cuda')
    assert L.nemo_rodrigues_fwd(M, th.data_ptr(), 0, R.data_ptr(), H.st()) == 0
    assert rel_err(R.reshape(-1, 3, 3), g['out']) < TOL
    d = torch.zeros(M, 3, device='cuda')
    assert L.nemo_rodrigues_bwd(M, th.data_ptr(), H.dev(g['ct']).data_ptr(), d.data_ptr(), H.st()) == 0
    assert rel_err(d, g['grad_theta']) < 1e-4
    g = load_golden('fn_lbs_rodrigues')
    th = H.dev(g['theta'])
    R = torch.zeros(th.shape[0], 9, device='cuda')
    assert L.nemo_rodrigues_fwd(th.shape[0], th.data_ptr(), 1, R.data_ptr(), H.st()) == 0
    assert rel_err(R.reshape(-1, 3, 3), g['out']) < TOL


# ------------------------------------------------------------------------------------------ phase / RBF
@pytest.mark.parametrize('kern', ['quadratic', 'linear', 'gaussian', 'inverse_quadratic', 'multiquadric',
                                  'inverse_multiquadric', 'spline', 'poisson_one', 'poisson_two', 'matern32',
                                  'matern52'])
def test_phase_embed_vs_oracle(L, kern):
    from oracle import ops
    from nemo_cvpr2023_amd.engine import RBF_KERNELS
    H = _ops()
    gen = torch.Generator().manual_seed(11)
    N, V, T, K, D, C = 37, 3, 9, 20, 16, 5
    vi = torch.randint(0, V, (N,), generator=gen)
    vi[vi == 1] = 0                                  # view 1 absent -> zero gradient rows
    fi = torch.randint(0, T, (N,), generator=gen)
    sh = (torch.linspace(0, 1, K).repeat(V, 1) + 0.05 * torch.randn(V, K, generator=gen))
    sc = 15 + 2 * torch.randn(V, K, generator=gen)
    sh[0, 3], sc[2, 5] = -0.1, -1.0                  # exercise the relu clamps
    ls = 0.3 * torch.randn(D, generator=gen)
    codes = torch.randn(V, C, generator=gen)
    ct = torch.randn(N + 1, D + C, generator=gen)
    # oracle
    sho, sco, lso, co = (t.clone().requires_grad_(True) for t in (sh, sc, ls, codes))
    raw = torch.linspace(0, 1, T)[fi].unsqueeze(1)
    ph = ops.monotonic_forward(sho[vi], sco[vi], raw)
    cen = torch.linspace(0, 1, D).unsqueeze(1)
    X = torch.cat([ops.rbf_forward(lso, cen, ph, kern), co[vi]], 1)
    X0 = torch.cat([ops.rbf_forward(lso, cen, torch.zeros(1, 1), kern), torch.zeros(1, C)], 1)
    Xo = torch.cat([X, X0], 0)
    (Xo * ct).sum().backward()
    # HIP: the V networks are stored interleaved [sh_0 | sc_0 | sh_1 | ...]
    pn = H.dev(torch.stack([sh, sc], 1).reshape(-1))
    Xd = torch.zeros(N + 1, D + C, device='cuda')
    phd, den = torch.zeros(N, device='cuda'), torch.zeros(N, device='cuda')
    dvi, dfi, dls, dco = H.dev(vi, torch.long), H.dev(fi, torch.long), H.dev(ls), H.dev(codes)
    kid = RBF_KERNELS[kern]
    xmeta = torch.zeros(64, device='cuda')
    assert L.nemo_phase_embed_fwd(N, V, T, K, D, C, dvi.data_ptr(), dfi.data_ptr(), None, pn.data_ptr(),
                                  pn.data_ptr() + 4 * K, 2 * K, dls.data_ptr(), dco.data_ptr(), None, kid,
                                  Xd.data_ptr(), D + C, phd.data_ptr(), den.data_ptr(), xmeta.data_ptr(), H.st()) == 0
    assert rel_err(phd, ph.detach().squeeze(1)) < 1e-5
    assert rel_err(Xd, Xo.detach()) < 1e-5
    assert H.meta_amax(xmeta) == float(Xd.abs().max())          # x_meta: the launch leaves max |X| in the scale record
    # the same launch with the step's zero-fills and Adam-table bookkeeping in further blocks (nemo_phase_embed_fwd_begin)
    from nemo_cvpr2023_amd._lib import AdamSeg
    import ctypes
    Xd2, phd2, den2 = torch.full_like(Xd, 7.0), torch.zeros_like(phd), torch.zeros_like(den)
    z0, z1 = torch.ones(1003, device='cuda'), torch.ones(77, device='cuda')           # (sizes not multiples of 16 bytes)
    seg_host = (AdamSeg * 2)()
    for i, (lr, st_) in enumerate(((1e-3, 4), (5e-2, 0))):
        seg_host[i].offset, seg_host[i].numel, seg_host[i].lr, seg_host[i].step = 0, 8, lr, st_
    seg_dev = torch.frombuffer(bytearray(bytes(seg_host)), dtype=torch.uint8).cuda()
    assert L.nemo_phase_embed_fwd_begin(N, V, T, K, D, C, dvi.data_ptr(), dfi.data_ptr(), None, pn.data_ptr(),
                                        pn.data_ptr() + 4 * K, 2 * K, dls.data_ptr(), dco.data_ptr(), None, kid,
                                        Xd2.data_ptr(), D + C, phd2.data_ptr(), den2.data_ptr(), None, z0.data_ptr(), 1003 * 4,
                                        z1.data_ptr(), 77 * 4, seg_dev.data_ptr(), 2, 0.9, 0.999, 0, None, H.st()) == 0
    assert torch.equal(Xd2, Xd) and torch.equal(phd2, phd) and torch.equal(den2, den)
    # ... and (ABI 18) with an absmax list in the launch's leading blocks: the records equal nemo_absmax_multi's, everything else as above
    from nemo_cvpr2023_amd import _lib
    g = torch.Generator().manual_seed(5)
    mats = [H.dev(torch.randn(1000, 1000, generator=g) * 3.0), H.dev(torch.randn(147, 1000, generator=g)),
            H.dev(torch.randn(1000, 105, generator=g) * 1e-3), H.dev(torch.randn(1, 1000, generator=g))]
    want = H.absmax_meta(*mats)
    for overwrite in (1, 0):
        got = torch.full((len(mats), 64), 5.0 if overwrite else 0.0, device='cuda')      # (overwrite: stale slots are replaced)
        d = (_lib.AbsmaxDesc * len(mats))()
        for i, x in enumerate(mats):
            d[i].src, d[i].rows, d[i].cols, d[i].lds, d[i].meta, d[i].overwrite = x.data_ptr(), x.shape[0], x.shape[1], x.stride(0), got[i].data_ptr(), overwrite
        Xd3, z0 = torch.full_like(Xd, 7.0), torch.ones(1003, device='cuda')
        assert L.nemo_phase_embed_fwd_begin(N, V, T, K, D, C, dvi.data_ptr(), dfi.data_ptr(), None, pn.data_ptr(),
                                            pn.data_ptr() + 4 * K, 2 * K, dls.data_ptr(), dco.data_ptr(), None, kid,
                                            Xd3.data_ptr(), D + C, phd2.data_ptr(), den2.data_ptr(), None, z0.data_ptr(), 1003 * 4,
                                            None, 0, None, 0, 0.9, 0.999, len(mats), d, H.st()) == 0
        assert torch.equal(Xd3, Xd) and float(z0.abs().sum()) == 0.0
        for i, x in enumerate(mats):
            assert H.meta_amax(got[i]) == float(x.abs().max()) == H.meta_amax(want[i])
    # a record inside a zero-filled range is refused (its blocks run beside the zero-fill's)
    d[0].meta = z0.data_ptr() + 64
    assert L.nemo_phase_embed_fwd_begin(N, V, T, K, D, C, dvi.data_ptr(), dfi.data_ptr(), None, pn.data_ptr(),
                                        pn.data_ptr() + 4 * K, 2 * K, dls.data_ptr(), dco.data_ptr(), None, kid,
                                        Xd3.data_ptr(), D + C, phd2.data_ptr(), den2.data_ptr(), None, z0.data_ptr(), 1003 * 4,
                                        None, 0, None, 0, 0.9, 0.999, len(mats), d, H.st()) != 0
    assert float(z0.abs().sum()) == 0.0 and float(z1.abs().sum()) == 0.0
    back = (AdamSeg * 2).from_buffer_copy(bytes(seg_dev.cpu().numpy().tobytes()))
    for i, (lr, st_) in enumerate(((1e-3, 4), (5e-2, 0))):
        assert back[i].step == st_ + 1
        assert abs(back[i].step_size - lr / (1 - 0.9 ** (st_ + 1))) <= 1e-6 * back[i].step_size
        assert abs(back[i].bias_corr2_sqrt - (1 - 0.999 ** (st_ + 1)) ** 0.5) <= 1e-6
    srt = int(bool((vi[1:] >= vi[:-1]).all()))          # the batch may be declared sorted by view only if it is
    for ws, hint in ((den, 0), (None, 0), (den, srt)):   # with the forward pass's denominators, and re-evaluating them
        gpn = torch.zeros_like(pn)
        gls, gco = torch.zeros(D, device='cuda'), torch.zeros(V, C, device='cuda')
        assert L.nemo_phase_embed_bwd(N, V, T, K, D, C, dvi.data_ptr(), dfi.data_ptr(), None, pn.data_ptr(),
                                      pn.data_ptr() + 4 * K, 2 * K, dls.data_ptr(), kid, phd.data_ptr(),
                                      H.dev(ct).data_ptr(), D + C, ws.data_ptr() if ws is not None else None,
                                      gpn.data_ptr(), gpn.data_ptr() + 4 * K,
                                      gls.data_ptr(), gco.data_ptr(), hint, H.st()) == 0
        gpn = gpn.reshape(V, 2, K)
        assert rel_err(gls, lso.grad) < 1e-4
        assert rel_err(gco, co.grad) < 1e-5
        assert rel_err(gpn[:, 0], sho.grad) < 1e-4
        assert rel_err(gpn[:, 1], sco.grad) < 1e-4
        assert float(gpn[1].abs().max()) == 0.0
    # the same backward with two batched bias column sums in further blocks of the launch (nemo_phase_embed_bwd_colsum)
    from nemo_cvpr2023_amd._lib import ColsumDesc
    Y1, Y2 = torch.randn(301, 70, generator=gen).cuda(), torch.randn(150, 200, generator=gen).cuda()
    o1, o2 = torch.ones(70, device='cuda'), torch.zeros(130, device='cuda')
    arr = (ColsumDesc * 2)()
    arr[0].X, arr[0].M, arr[0].N, arr[0].ldx, arr[0].out = Y1.data_ptr(), 301, 70, 70, o1.data_ptr()
    arr[1].X, arr[1].M, arr[1].N, arr[1].ldx, arr[1].out = Y2.data_ptr(), 150, 130, 200, o2.data_ptr()
    gpn2 = torch.zeros_like(pn)
    gls2, gco2 = torch.zeros(D, device='cuda'), torch.zeros(V, C, device='cuda')
    assert L.nemo_phase_embed_bwd_colsum(N, V, T, K, D, C, dvi.data_ptr(), dfi.data_ptr(), None, pn.data_ptr(),
                                         pn.data_ptr() + 4 * K, 2 * K, dls.data_ptr(), kid, phd.data_ptr(),
                                         H.dev(ct).data_ptr(), D + C, den.data_ptr(), gpn2.data_ptr(), gpn2.data_ptr() + 4 * K,
                                         gls2.data_ptr(), gco2.data_ptr(), 2, arr, 0, H.st()) == 0
    assert rel_err(gpn2.reshape(V, 2, K), gpn) < 1e-5 and rel_err(gls2, gls) < 1e-5 and rel_err(gco2, gco) < 1e-5
    assert rel_err(o1, 1.0 + Y1.double().sum(0)) < 1e-5 and rel_err(o2, Y2[:, :130].double().sum(0)) < 1e-5


# ------------------------------------------------------------------------------------------ SMPL pieces
def _ctx(num_verts, version=2, skin_nnz=24):
    from nemo_cvpr2023_amd.engine import SmplContext
    assets = syn.make_smpl_assets(num_verts, seed=1, skin_nnz=skin_nnz)
    jm = [int(x) for x in assets['joint_map']]
    idx = list(range(25)) if version == 4 else [38] + list(range(1, 25))
    return assets, SmplContext(assets, [jm[i] for i in idx], 'cuda:0'), idx


def _rand_rot(gen, n, small=False):
    from oracle import ops
    th = torch.randn(n, 3, generator=gen) * (0.3 if small else 1.5)
    return ops.batch_rodrigues(th)


@pytest.mark.parametrize('order', ['random', 'sorted_by_view'])
@pytest.mark.parametrize('num_verts,version', [(128, 2), (128, 4), (6890, 2)])
def test_keypoint_path_forward_backward(L, num_verts, version, order):
    """FK -> pre-contracted mesh joints -> projection -> robust loss, and the whole adjoint chain,
    against the oracle's unfused lbs + autograd.  order: the per-view sums (round 5: deposits + one block per view) take a
    view's samples from a scan of the batch, or -- batch sorted by view, here with a view that has no sample -- from the view
    bounds the depositing lanes leave behind."""
    from oracle import ops
    H = _ops()
    assets, ctx, idx = _ctx(num_verts, version)
    gen = torch.Generator().manual_seed(21)
    N, V, T = 13, 3, 6
    R = _rand_rot(gen, N * 24, small=True).reshape(N, 24, 3, 3)
    TR = 0.2 * torch.randn(N + 1, 3, generator=gen)
    cams = 1e-2 * torch.randn(V, 9, generator=gen)
    cams[:, 3] += 1; cams[:, 6] += 1; cams[:, 2] += 9.26
    vi = torch.randint(0, V, (N,), generator=gen); vi[vi == 1] = 2
    fi = torch.randint(0, T, (N,), generator=gen)
    if order == 'sorted_by_view':
        vi, perm = vi.sort()
        fi = fi[perm]
    seqs = syn.SyntheticSequences(V, T, seed=5)
    tgt = torch.tensor(np.array([np.array(s['pose_2d_op']) for s in seqs.sequences]))
    # oracle
    Ro, TRo, co = (t.clone().requires_grad_(True) for t in (R, TR, cams))
    smpl = ops.SMPLOracle(assets)
    _, j49, _ = smpl.forward(torch.zeros(1, 10), Ro)
    j = j49[:, idx] + (TRo[:N] - TRo[N:]).unsqueeze(1)
    Rc = ops.rot6d_to_rotmat(co[vi][:, 3:])
    cen = torch.tensor([[540.0, 960.0]]).expand(N, -1)
    p2 = ops.perspective_projection(j, Rc, co[vi][:, :3], 5000.0, cen)
    gt = tgt[vi, fi]
    la = ops.keypoint_loss(p2, gt[..., :2], gt[..., 2:], None, 'mse_robust')
    kp = ops.per_view_mean_loss(la, gt[..., -1:], vi)
    kp.backward()
    # HIP
    dR, dTR, dc = H.dev(R.reshape(N, 24, 9)), H.dev(TR), H.dev(cams)
    dvi, dfi, dt = H.dev(vi, torch.long), H.dev(fi, torch.long), H.dev(tgt)
    Z = lambda *s: torch.zeros(*s, device='cuda')
    A, Jp, PF = Z(N, 24, 12), Z(N, 24, 3), Z(N, 207)
    assert L.nemo_fk_fwd(ctx.handle, N, dR.data_ptr(), A.data_ptr(), Jp.data_ptr(), PF.data_ptr(), 207, H.st()) == 0
    nq72 = ctx.nq * 72
    Mq = Z(N, nq72)
    assert L.nemo_gemm_f32(0, 0, N, nq72, 207, PF.data_ptr(), 207, ctx.C1, nq72, Mq.data_ptr(), nq72, ctx.c0, 0,
                           None, 0, 0, 1.0, 0, 1, None, 0, H.st()) == 0
    j3d, p2d, lall, vacc, norm, scal = Z(N, 25, 3), Z(N, 25, 2), Z(N, 25, 2), Z(V, 2), Z(1), Z(8)
    args = (ctx.handle, N, V, T, A.data_ptr(), Jp.data_ptr(), Mq.data_ptr(), nq72, dTR.data_ptr(), 3, 1,
            dvi.data_ptr(), dfi.data_ptr(), dc.data_ptr(), dt.data_ptr(), None, 5000.0, 540.0, 960.0, 0, 0)
    assert L.nemo_kp_fwd(*args, j3d.data_ptr(), p2d.data_ptr(), lall.data_ptr(), vacc.data_ptr(), None, H.st()) == 0
    assert L.nemo_kp_finalize(V, 25, 2, 0, vacc.data_ptr(), scal.data_ptr(), norm.data_ptr(), H.st()) == 0
    assert rel_err(j3d, j.detach()) < 1e-5
    assert rel_err(p2d, p2.detach()) < 1e-5
    assert rel_err(lall, la.detach()) < 1e-4
    assert rel_err(scal[0], kp.detach()) < 1e-5
    assert float(norm[0]) == 2.0
    dA, dJp, dMq, dTRg, dcg, dPF, dRg = (Z(N, 24, 12), Z(N, 24, 3), Z(N, nq72), Z(N + 1, 3), Z(V, 9), Z(N, 207),
                                         Z(N, 24, 9))
    assert L.nemo_kp_bwd(*args, vacc.data_ptr(), norm.data_ptr(), 1.0, dA.data_ptr(), dJp.data_ptr(),
                         dMq.data_ptr(), dTRg.data_ptr(), 3, dcg.data_ptr(), H.st()) == 0
    assert L.nemo_gemm_f32(0, 1, N, 207, nq72, dMq.data_ptr(), nq72, ctx.C1, nq72, dPF.data_ptr(), 207, None, 0,
                           None, 0, 0, 1.0, 0, 1, None, 0, H.st()) == 0
    assert L.nemo_fk_bwd(ctx.handle, N, dR.data_ptr(), A.data_ptr(), dA.data_ptr(), dJp.data_ptr(),
                         dPF.data_ptr(), 207, dRg.data_ptr(), H.st()) == 0
    assert L.nemo_scale_neg_rowsum(N, 3, dTRg.data_ptr(), 3, dTRg.data_ptr() + 4 * 3 * N, H.st()) == 0
    assert rel_err(dRg.reshape(N, 24, 3, 3), Ro.grad) < 1e-4
    assert rel_err(dTRg, TRo.grad) < 1e-4
    assert rel_err(dcg, co.grad) < 1e-4
    assert float(dcg[1].abs().max()) == 0.0
    # norm == NULL: nemo_kp_bwd derives the normaliser from the view accumulators itself (the step runs kp_finalize off
    # the main chain) -- bit-identical outputs
    dA2, dJp2, dMq2, dTR2, dc2 = Z(N, 24, 12), Z(N, 24, 3), Z(N, nq72), Z(N + 1, 3), Z(V, 9)
    assert L.nemo_kp_bwd(*args, vacc.data_ptr(), None, 1.0, dA2.data_ptr(), dJp2.data_ptr(), dMq2.data_ptr(),
                         dTR2.data_ptr(), 3, dc2.data_ptr(), H.st()) == 0
    assert torch.equal(dA2, dA) and torch.equal(dMq2, dMq) and torch.equal(dTR2[:N], dTRg[:N])
    assert rel_err(dJp2, dJp) < 1e-6 and rel_err(dc2, dcg) < 1e-6          # (atomic accumulations: order may differ)
    # nemo_kp_fwd_bwd (round 4): forward and backward in ONE launch, normalised with the per-view sample counts the caller knows
    # from the indices -- every output of both launches, bit for bit where no atomics are involved
    cnt = H.dev(torch.bincount(vi, minlength=V), torch.long)
    j3, p3, la3, va3 = Z(N, 25, 3), Z(N, 25, 2), Z(N, 25, 2), Z(V, 2)
    dA3, dJp3, dMq3, dTR3, dc3 = Z(N, 24, 12), Z(N, 24, 3), Z(N, nq72), Z(N + 1, 3), Z(V, 9)
    assert L.nemo_kp_fwd_bwd(*args, cnt.data_ptr(), 1.0, j3.data_ptr(), p3.data_ptr(), la3.data_ptr(), va3.data_ptr(),
                             dA3.data_ptr(), dJp3.data_ptr(), dMq3.data_ptr(), dTR3.data_ptr(), 3, dc3.data_ptr(), None,
                             H.st()) == 0
    assert torch.equal(j3, j3d) and torch.equal(p3, p2d) and torch.equal(la3, lall)
    assert torch.equal(va3[:, 1], vacc[:, 1]) and rel_err(va3[:, 0], vacc[:, 0]) < 1e-6
    assert torch.equal(dA3, dA) and torch.equal(dMq3, dMq) and torch.equal(dTR3[:N], dTRg[:N])
    assert rel_err(dJp3, dJp) < 1e-6 and rel_err(dc3, dcg) < 1e-6
    # ... and with padding rows (n_valid): no loss, not counted, exactly-zero gradient rows -- as the two launches do it
    nv = H.dev(torch.tensor([N - 4]), torch.long)
    cntp = H.dev(torch.bincount(vi[:N - 4], minlength=V), torch.long)
    va4, dA4, dJp4, dMq4, dTR4, dc4 = Z(V, 2), Z(N, 24, 12), Z(N, 24, 3), Z(N, nq72), Z(N + 1, 3), Z(V, 9)
    vaP, dAP, dJpP, dMqP, dTRP, dcP = Z(V, 2), Z(N, 24, 12), Z(N, 24, 3), Z(N, nq72), Z(N + 1, 3), Z(V, 9)
    assert L.nemo_kp_fwd(*args, j3d.data_ptr(), p2d.data_ptr(), lall.data_ptr(), vaP.data_ptr(), nv.data_ptr(), H.st()) == 0
    assert L.nemo_kp_bwd_ex(*args, vaP.data_ptr(), None, 1.0, dAP.data_ptr(), dJpP.data_ptr(), dMqP.data_ptr(),
                            dTRP.data_ptr(), 3, dcP.data_ptr(), None, nv.data_ptr(), H.st()) == 0
    assert L.nemo_kp_fwd_bwd(*args, cntp.data_ptr(), 1.0, j3.data_ptr(), p3.data_ptr(), la3.data_ptr(), va4.data_ptr(),
                             dA4.data_ptr(), dJp4.data_ptr(), dMq4.data_ptr(), dTR4.data_ptr(), 3, dc4.data_ptr(),
                             nv.data_ptr(), H.st()) == 0
    assert torch.equal(la3, lall) and torch.equal(va4[:, 1], vaP[:, 1]) and rel_err(va4[:, 0], vaP[:, 0]) < 1e-6
    assert torch.equal(dA4, dAP) and torch.equal(dMq4, dMqP) and torch.equal(dTR4[:N], dTRP[:N])
    assert float(dA4[N - 4:].abs().max()) == 0.0 and float(dMq4[N - 4:].abs().max()) == 0.0
    assert rel_err(dJp4, dJpP) < 1e-6 and rel_err(dc4, dcP) < 1e-6


@pytest.mark.parametrize('loss_type,lid', [('mse', 1), ('rmse', 2), ('rmse_robust', 3),
                                           ('mse_robust_resized', 4), ('rmse_resized', 5)])
def test_keypoint_loss_types_and_camera_mode(L, loss_type, lid):
    from oracle import ops
    H = _ops()
    assets, ctx, idx = _ctx(128, 2)
    gen = torch.Generator().manual_seed(31 + lid)
    N, V, T = 9, 2, 5
    R = _rand_rot(gen, N * 24, small=True).reshape(N, 24, 3, 3)
    cams = 1e-2 * torch.randn(V, 9, generator=gen)
    cams[:, 3] += 1; cams[:, 6] += 1; cams[:, 2] += 9.26
    vi, fi = torch.randint(0, V, (N,), generator=gen), torch.randint(0, T, (N,), generator=gen)
    seqs = syn.SyntheticSequences(V, T, seed=6)
    tgt = torch.tensor(np.array([np.array(s['pose_2d_op']) for s in seqs.sequences]))
    size = 200 + 500 * torch.rand(V, T, generator=gen)
    co = cams.clone().requires_grad_(True)
    smpl = ops.SMPLOracle(assets)
    _, j49, _ = smpl.forward(torch.zeros(1, 10), R)
    p2 = ops.perspective_projection(j49[:, idx], ops.rot6d_to_rotmat(co[vi][:, 3:]), co[vi][:, :3], 5000.0,
                                    torch.tensor([[540.0, 960.0]]).expand(N, -1))
    gt = tgt[vi, fi]
    la = ops.keypoint_loss(p2, gt[..., :2], gt[..., 2:], size[vi, fi], loss_type)
    mean = la.mean()                      # camera_fitting_loss (mean_mode 1)
    mean.backward()
    Z = lambda *s: torch.zeros(*s, device='cuda')
    dR, dc = H.dev(R.reshape(N, 24, 9)), H.dev(cams)
    dvi, dfi, dt, dsz = H.dev(vi, torch.long), H.dev(fi, torch.long), H.dev(tgt), H.dev(size)
    A, Jp, PF = Z(N, 24, 12), Z(N, 24, 3), Z(N, 207)
    L.nemo_fk_fwd(ctx.handle, N, dR.data_ptr(), A.data_ptr(), Jp.data_ptr(), PF.data_ptr(), 207, H.st())
    nq72 = ctx.nq * 72
    Mq = Z(N, nq72)
    L.nemo_gemm_f32(0, 0, N, nq72, 207, PF.data_ptr(), 207, ctx.C1, nq72, Mq.data_ptr(), nq72, ctx.c0, 0, None,
                    0, 0, 1.0, 0, 1, None, 0, H.st())
    W = la.shape[-1]
    lall, vacc, norm, scal, dcg = Z(N, 25, W), Z(V, 2), Z(1), Z(8), Z(V, 9)
    args = (ctx.handle, N, V, T, A.data_ptr(), Jp.data_ptr(), Mq.data_ptr(), nq72, None, 3, 0, dvi.data_ptr(),
            dfi.data_ptr(), dc.data_ptr(), dt.data_ptr(), dsz.data_ptr(), 5000.0, 540.0, 960.0, lid, 1)
    assert L.nemo_kp_fwd(*args, None, None, lall.data_ptr(), vacc.data_ptr(), None, H.st()) == 0
    assert L.nemo_kp_finalize(V, 25, W, 1, vacc.data_ptr(), scal.data_ptr(), norm.data_ptr(), H.st()) == 0
    assert rel_err(lall, la.detach()) < 1e-4
    assert rel_err(scal[0], mean.detach()) < 1e-5
    assert L.nemo_kp_bwd(*args, vacc.data_ptr(), norm.data_ptr(), 1.0, None, None, None, None, 3,
                         dcg.data_ptr(), H.st()) == 0
    assert rel_err(dcg, co.grad) < 1e-4


def _l1_grad_gate(assets, R2, N, dRg, Ro_grad):
    """dRg (HIP) against the oracle's autograd gradient of sum |v_rec - v_orig| w.r.t. the N x 24 rotations: 1e-4 of the largest
    entry + 3 x the fp32 oracle's own distance from a float64 evaluation + the elementwise bound of what sign(0) ties can change
    (tests/tiebound.py) -- instead of the flat 2e-3 of rounds 1 - 4."""
    from oracle import ops
    from tiebound import l1_tie_bound
    a64 = {k: (v.double() if isinstance(v, torch.Tensor) and v.is_floating_point() else v) for k, v in assets.items()}
    smpl64 = ops.SMPLOracle(a64)
    Ro64 = R2[:N].double().clone().requires_grad_(True)
    vr64 = smpl64.forward(torch.zeros(1, 10, dtype=torch.float64), R2[N:].double())[0].detach()
    (bound,), n_ties = l1_tie_bound(lambda: smpl64.forward(torch.zeros(1, 10, dtype=torch.float64), Ro64)[0], [Ro64], vr64, 1.0)
    vo64 = smpl64.forward(torch.zeros(1, 10, dtype=torch.float64), Ro64)[0]
    (g64,) = torch.autograd.grad((vr64 - vo64).abs().sum(), [Ro64])
    go = Ro_grad.double()
    gh = dRg.detach().cpu().double().reshape(go.shape)
    scale, noise = float(go.abs().max()), float((go - g64).abs().max())
    diff = (gh - go).abs()
    assert bool((diff <= 1e-4 * scale + 3.0 * noise + bound).all()), (float(diff.max()), scale, noise, float(bound.max()), n_ties)


@pytest.mark.parametrize('num_verts', [128, 6890])
def test_vertices_and_v2v(L, num_verts):
    """Full-mesh path: pose-blend GEMM + skinning; the fused L1 + gradient kernel against autograd."""
    from oracle import ops
    H = _ops()
    assets, ctx, idx = _ctx(num_verts, 2)
    gen = torch.Generator().manual_seed(41)
    N = 6
    NV3 = 3 * num_verts
    R2 = _rand_rot(gen, 2 * N * 24, small=True).reshape(2 * N, 24, 3, 3)
    R2[N:, 0] = R2[:N, 0]
    Ro = R2[:N].clone().requires_grad_(True)
    smpl = ops.SMPLOracle(assets)
    vo, _, _ = smpl.forward(torch.zeros(1, 10), Ro)
    vr, _, _ = smpl.forward(torch.zeros(1, 10), R2[N:])
    l1 = (vr.detach() - vo).abs().sum()
    l1.backward()
    Z = lambda *s: torch.zeros(*s, device='cuda')
    dR2 = H.dev(R2.reshape(2 * N, 24, 9))
    ldP = ctx.ldP
    assert ldP % 4 == 0 and ldP == 3 * ctx.NVp and 0 <= ctx.NVp - num_verts < 16
    A, Jp, PF, VP = Z(2 * N, 24, 12), Z(2 * N, 24, 3), Z(2 * N, 207), Z(2 * N, NV3)
    assert L.nemo_fk_fwd(ctx.handle, 2 * N, dR2.data_ptr(), A.data_ptr(), Jp.data_ptr(), PF.data_ptr(), 207,
                         H.st()) == 0
    assert L.nemo_gemm_f32(0, 0, 2 * N, NV3, 207, PF.data_ptr(), 207, ctx.posedirs, ldP, VP.data_ptr(), NV3,
                           ctx.v_shaped, 0, None, 0, 0, 1.0, 0, 1, None, 0, H.st()) == 0
    verts = Z(2 * N, num_verts, 3)
    tr = H.dev(0.1 * torch.randn(2 * N, 3, generator=gen))
    assert L.nemo_skin_vertices(ctx.handle, 2 * N, VP.data_ptr(), NV3, A.data_ptr(), tr.data_ptr(), 3,
                                verts.data_ptr(), H.st()) == 0
    ref = torch.cat([vo.detach(), vr], 0) + tr.cpu().unsqueeze(1)
    assert rel_err(verts, ref) < 1e-5
    loss, dVP, dA, dPF, dRg = Z(1), Z(N, NV3), Z(N, 24, 12), Z(N, 207), Z(N, 24, 9)
    assert L.nemo_v2v_skin_l1(ctx.handle, N, VP.data_ptr(), NV3, A.data_ptr(), loss.data_ptr(), dVP.data_ptr(),
                              NV3, dA.data_ptr(), H.st()) == 0
    assert rel_err(loss[0], l1.detach()) < 1e-5
    assert L.nemo_gemm_f32(0, 1, N, 207, NV3, dVP.data_ptr(), NV3, ctx.posedirs, ldP, dPF.data_ptr(), 207, None,
                           0, None, 0, 0, 1.0, 2, 8, None, 0, H.st()) == 0
    assert L.nemo_fk_bwd(ctx.handle, N, dR2.data_ptr(), A.data_ptr(), dA.data_ptr(), None, dPF.data_ptr(), 207,
                         dRg.data_ptr(), H.st()) == 0
    # |.| is non-smooth: a vertex coordinate within rounding of a tie may flip a sign -- bounded analytically, not by a flat gate
    _l1_grad_gate(assets, R2, N, dRg, Ro.grad)


@pytest.mark.parametrize('num_verts,N,plan', [(128, 6, ''), (100, 37, ''), (6890, 20, ''),
                                              # forced grid plans "RA,Lr,k" (the shapes N = 600 ... 6000 get
                                              # by themselves): left-over blocks over 2 / 3 sample groups
                                              (6890, 40, '3,9,3'), (6890, 40, '1,36,2'), (100, 37, '1,1,2'),
                                              (128, 50, '1,1,3'), (6890, 20, '5,0,0')])
@pytest.mark.parametrize('skin_nnz', [24, 4, 3])
@pytest.mark.parametrize('entry', ['nemo_v2v_fused', 'nemo_v2v_fused_split'])
def test_v2v_fused_mesh_kernel(L, num_verts, N, plan, skin_nnz, entry, monkeypatch):
    """Fused pose blend + skinning + L1 + gradient (MFMA accumulator layout end to end) against the
    oracle's unfused lbs + autograd; ragged vertex tiles (100, 6890 = 430*16+10) and sample groups.
    skin_nnz: 24 = dense weights (24-joint skinning product on the MFMA pipe), 4 / 3 = the published model's sparsity
    (skinning with the non-zero weights only, csrc/smpl.hip SPARSE).  entry: the fp32-MFMA kernel (MODE 0) and the
    fp32-equivalent three-piece form the engine runs by default (MODE 4) -- same gates."""
    fused = getattr(L, entry)
    if plan:
        monkeypatch.setenv('NEMO_MESH_PLAN', plan)
    from oracle import ops
    H = _ops()
    assets, ctx, idx = _ctx(num_verts, 2, skin_nnz)
    assert ctx.skin_nnz == skin_nnz and ctx.skin_sparse == (skin_nnz <= 4)
    gen = torch.Generator().manual_seed(47 + N)
    NV3 = 3 * num_verts
    R2 = _rand_rot(gen, 2 * N * 24, small=True).reshape(2 * N, 24, 3, 3)
    R2[N:, 0] = R2[:N, 0]
    Ro = R2[:N].clone().requires_grad_(True)
    smpl = ops.SMPLOracle(assets)
    vo, _, _ = smpl.forward(torch.zeros(1, 10), Ro)
    vr, _, _ = smpl.forward(torch.zeros(1, 10), R2[N:])
    l1 = (vr.detach() - vo).abs().sum()
    l1.backward()
    Z = lambda *s: torch.zeros(*s, device='cuda')
    dR2 = H.dev(R2.reshape(2 * N, 24, 9))
    A, Jp, PF = Z(2 * N, 24, 12), Z(2 * N, 24, 3), Z(2 * N, 208)
    assert L.nemo_fk_fwd(ctx.handle, 2 * N, dR2.data_ptr(), A.data_ptr(), Jp.data_ptr(), PF.data_ptr(), 208,
                         H.st()) == 0
    ldn = (N + 15) // 16 * 16
    loss, dVPt, dA, dPF, dRg = Z(1), Z(3 * ctx.NVp, ldn), Z(N, 24, 12), Z(N, 208), Z(N, 24, 9)
    ws = torch.zeros(int(L.nemo_v2v_fused_ws_bytes(ctx.handle, N)) // 4 + 1, device='cuda')
    dA.fill_(7.0)                                       # dA is overwritten, not accumulated
    for _ in range(2):                                  # second launch: tickets were returned to zero
        loss.zero_()
        assert fused(ctx.handle, N, PF.data_ptr(), 208, A.data_ptr(), loss.data_ptr(), dVPt.data_ptr(),
                                ldn, dA.data_ptr(), ws.data_ptr(), ws.numel() * 4, H.st()) == 0
    assert rel_err(loss[0], l1.detach()) < 1e-5
    assert float(dVPt[NV3:].abs().sum()) == 0.0 and float(dVPt[:, N:].abs().sum()) == 0.0   # pads stay zero
    # deferred combine (dA = NULL + nemo_v2v_combine, what the step runs beside the adjoint GEMM): bit-identical dA, and
    # an in-launch combine afterwards still finds its tickets at zero
    loss2, dVPt2, dA2 = Z(1), Z(3 * ctx.NVp, ldn), Z(N, 24, 12).fill_(-3.0)
    assert fused(ctx.handle, N, PF.data_ptr(), 208, A.data_ptr(), loss2.data_ptr(), dVPt2.data_ptr(),
                            ldn, None, ws.data_ptr(), ws.numel() * 4, H.st()) == 0
    assert L.nemo_v2v_combine(ctx.handle, N, dA2.data_ptr(), ws.data_ptr(), ws.numel() * 4, H.st()) == 0
    assert torch.equal(dA2, dA) and torch.equal(dVPt2, dVPt) and torch.equal(loss2, loss)
    dA3 = Z(N, 24, 12)
    loss2.zero_()
    assert fused(ctx.handle, N, PF.data_ptr(), 208, A.data_ptr(), loss2.data_ptr(), dVPt2.data_ptr(),
                            ldn, dA3.data_ptr(), ws.data_ptr(), ws.numel() * 4, H.st()) == 0
    assert torch.equal(dA3, dA)
    assert L.nemo_gemm_f32(1, 1, N, 207, NV3, dVPt.data_ptr(), ldn, ctx.posedirs, ctx.ldP, dPF.data_ptr(), 208,
                           None, 0, None, 0, 0, 1.0, 2, 8, None, 0, H.st()) == 0
    assert L.nemo_fk_bwd(ctx.handle, N, dR2.data_ptr(), A.data_ptr(), dA.data_ptr(), None, dPF.data_ptr(), 208,
                         dRg.data_ptr(), H.st()) == 0
    # |.| is non-smooth: a coordinate within rounding of a tie may flip a sign -- bounded analytically, not by a flat gate
    _l1_grad_gate(assets, R2, N, dRg, Ro.grad)


@pytest.mark.parametrize('bf16', [False, True])
@pytest.mark.parametrize('num_verts,N', [(6890, 40), (100, 37)])
def test_sparse_skinning_is_the_dense_product_without_its_zero_terms(L, num_verts, N, bf16):
    """On 4-sparse weights the sparse form of the fused mesh kernel and its dense 24-joint form (the switch is
    nemo_ctx_set_skin_sparse) add the same non-zero terms in the same (ascending joint) order: loss, d vp and dA are
    bit-identical.  A model with five non-zero weights on some vertex keeps the dense form and refuses the sparse one."""
    H = _ops()
    assets, ctx, idx = _ctx(num_verts, 2, 4)
    gen = torch.Generator().manual_seed(5)
    R2 = H.dev(_rand_rot(gen, 2 * N * 24, small=True).reshape(2 * N, 24, 9))
    Z = lambda *s: torch.zeros(*s, device='cuda')
    A, Jp, PF = Z(2 * N, 24, 12), Z(2 * N, 24, 3), Z(2 * N, 208)
    assert L.nemo_fk_fwd(ctx.handle, 2 * N, R2.data_ptr(), A.data_ptr(), Jp.data_ptr(), PF.data_ptr(), 208, H.st()) == 0
    ldn = (N + 15) // 16 * 16
    ws = torch.zeros(int(L.nemo_v2v_fused_ws_bytes(ctx.handle, N)) // 4 + 1, device='cuda')
    fn = L.nemo_v2v_fused_bf16 if bf16 else L.nemo_v2v_fused
    out = {}
    for sparse in (True, False):
        ctx.set_skin_sparse(sparse)
        assert ctx.skin_sparse == sparse
        loss, dVPt, dA = Z(1), Z(3 * ctx.NVp, ldn), Z(N, 24, 12)
        assert fn(ctx.handle, N, PF.data_ptr(), 208, A.data_ptr(), loss.data_ptr(), dVPt.data_ptr(), ldn, dA.data_ptr(),
                  ws.data_ptr(), ws.numel() * 4, H.st()) == 0
        torch.cuda.synchronize()
        out[sparse] = (loss, dVPt, dA)
    for a, b in zip(out[True], out[False]):
        assert torch.equal(a, b)
    assert float(out[True][0]) > 0 and float(out[True][2].abs().sum()) > 0
    a5, c5, _ = _ctx(num_verts, 2, 5)
    assert c5.skin_nnz == 5 and not c5.skin_sparse
    with pytest.raises(RuntimeError):
        c5.set_skin_sparse(True)


def test_v2v_prep(L):
    from oracle import ops
    H = _ops()
    gen = torch.Generator().manual_seed(43)
    N = 5
    R = _rand_rot(gen, N * 24).reshape(N, 24, 9)
    aa, aad = 0.5 * torch.randn(N, 72, generator=gen), 0.5 * torch.randn(N, 63, generator=gen)
    R2 = torch.zeros(2 * N, 24, 9, device='cuda')
    assert L.nemo_v2v_prep_fwd(N, H.dev(R).data_ptr(), H.dev(aa).data_ptr(), H.dev(aad).data_ptr(),
                               R2.data_ptr(), None, H.st()) == 0
    aao = aa.clone().requires_grad_(True)
    Ro = ops.batch_rodrigues(aao[:, 3:].reshape(-1, 3)).reshape(N, 23, 9)
    Rr = ops.batch_rodrigues(torch.cat([aad, aa[:, 66:]], 1).reshape(-1, 3)).reshape(N, 23, 9)
    assert rel_err(R2[:N, 1:], Ro.detach()) < TOL and rel_err(R2[N:, 1:], Rr) < TOL
    assert rel_err(R2[:N, 0], R[:, 0]) == 0 and rel_err(R2[N:, 0], R[:, 0]) == 0
    ct = torch.randn(N, 24, 9, generator=gen)
    (Ro * ct[:, 1:]).sum().backward()
    daa, dRm = torch.zeros(N, 72, device='cuda'), torch.zeros(N, 24, 9, device='cuda')
    assert L.nemo_v2v_prep_bwd(N, H.dev(aa).data_ptr(), H.dev(ct).data_ptr(), 0.5, daa.data_ptr(),
                               dRm.data_ptr(), H.st()) == 0
    assert rel_err(daa, 0.5 * aao.grad) < 1e-4
    assert rel_err(dRm[:, 0], 0.5 * ct[:, 0]) < TOL and float(dRm[:, 1:].abs().max()) == 0.0
    # the decoder's 6-D output converted inside the same launch (nemo_v2v_prep_fwd_dec) == nemo_rot6d_fwd + nemo_v2v_prep_fwd,
    # with and without padding rows
    d6 = H.dev(torch.randn(N, 130, generator=gen))                          # (row stride 130 >= 126)
    aad2 = torch.zeros(N, 63, device='cuda')
    assert L.nemo_rot6d_fwd(N, 21, d6.data_ptr(), 130, 0, None, aad2.data_ptr(), H.st()) == 0
    for nv in (None, torch.tensor([3], device='cuda')):
        Ra, Rb = torch.zeros(2 * N, 24, 9, device='cuda'), torch.zeros(2 * N, 24, 9, device='cuda')
        aad3 = torch.zeros(N, 63, device='cuda')
        nvp = nv.data_ptr() if nv is not None else None
        assert L.nemo_v2v_prep_fwd(N, H.dev(R).data_ptr(), H.dev(aa).data_ptr(), aad2.data_ptr(), Ra.data_ptr(), nvp,
                                   H.st()) == 0
        assert L.nemo_v2v_prep_fwd_dec(N, H.dev(R).data_ptr(), H.dev(aa).data_ptr(), d6.data_ptr(), 130, aad3.data_ptr(),
                                       Rb.data_ptr(), nvp, H.st()) == 0
        assert rel_err(aad3, aad2) < 1e-6 and rel_err(Rb, Ra) < 1e-6
        if nv is not None:
            assert torch.equal(Rb[N + 3:], Rb[3:N])                        # padding rows: the first body twice


# ------------------------------------------------------------------------------------------ priors
def test_kl_gmm_pose3d(L):
    from oracle import ops
    from nemo_cvpr2023_amd.engine import gmm_constants
    H = _ops()
    gen = torch.Generator().manual_seed(51)
    N = 70
    mulv = torch.randn(N, 64, generator=gen)
    mulv[0, 40] = 25.0                                 # softplus threshold branch
    mo = mulv.clone().requires_grad_(True)
    kl = ops.kl_to_std_normal(mo[:, :32], torch.nn.functional.softplus(mo[:, 32:]))
    kl.backward()
    out, d = torch.zeros(8, device='cuda'), torch.zeros(N, 64, device='cuda')
    assert L.nemo_kl_fwd_bwd(N, 32, H.dev(mulv).data_ptr(), 64, out.data_ptr(), d.data_ptr(), 64, None, H.st()) == 0
    assert rel_err(out[0], kl.detach()) < 1e-5 and rel_err(d, mo.grad) < 1e-4

    g = syn.make_gmm()
    prior = ops.GMMPriorOracle(g)
    c = gmm_constants(g, 'cuda:0')
    x = 0.3 * torch.randn(N, 72, generator=gen)
    xo = x.clone().requires_grad_(True)
    ll = prior(xo[:, 3:])
    ll.mean().backward()
    per, dx = torch.zeros(N, device='cuda'), torch.zeros(N, 72, device='cuda')
    dxp = H.dev(x)
    assert L.nemo_gmm_fwd_bwd(N, 8, 69, dxp.data_ptr() + 12, 72, c['means'].data_ptr(), c['prec'].data_ptr(),
                              c['log_nllw'].data_ptr(), torch.zeros(N, 8, device='cuda').data_ptr(),
                              out.data_ptr() + 4, per.data_ptr(), 2.0,
                              dx.data_ptr() + 12, 72, None, H.st()) == 0
    assert rel_err(per, ll.detach()) < 1e-5 and rel_err(out[1], ll.mean().detach()) < 1e-5
    assert rel_err(dx, 2.0 * xo.grad) < 1e-4

    V, T = 3, 4
    theta, mask = 0.2 * torch.randn(V, T, 69, generator=gen), (torch.rand(V, T, 1, generator=gen) > 0.3).float()
    vi, fi = torch.randint(0, V, (N,), generator=gen), torch.randint(0, T, (N,), generator=gen)
    xo = x.clone().requires_grad_(True)
    l3 = ops.keypoint_loss(xo[:, 3:], theta[vi, fi], mask[vi, fi], None, 'mse_robust').mean()
    l3.backward()
    dx.zero_()
    assert L.nemo_pose3d_fwd_bwd(N, 69, dxp.data_ptr() + 12, 72, H.dev(theta).data_ptr(), H.dev(mask).data_ptr(),
                                 H.dev(vi, torch.long).data_ptr(), H.dev(fi, torch.long).data_ptr(), T,
                                 out.data_ptr() + 8, 1.0, dx.data_ptr() + 12, 72, None, H.st()) == 0
    assert rel_err(out[2], l3.detach()) < 1e-5 and rel_err(dx, xo.grad) < 1e-4


@pytest.mark.parametrize('adamw,wd', [(False, 0.0), (False, 0.001), (True, 0.01)])
def test_fused_adam_matches_torch(L, adamw, wd):
    from nemo_cvpr2023_amd._lib import AdamSeg
    H = _ops()
    gen = torch.Generator().manual_seed(61)
    n = 5000
    p0 = torch.randn(n, generator=gen)
    pt = p0.clone().requires_grad_(True)
    opt = (torch.optim.AdamW if adamw else torch.optim.Adam)([pt], lr=1e-2, weight_decay=wd)
    p, m, v = H.dev(p0), torch.zeros(n, device='cuda'), torch.zeros(n, device='cuda')
    for t in range(1, 6):
        gr = torch.randn(n, generator=gen) * (10.0 ** -t)
        pt.grad = gr.clone()
        opt.step()
        seg = (AdamSeg * 2)()
        for i, (a, b) in enumerate(((0, 2000), (2000, 3000))):
            seg[i].offset, seg[i].numel, seg[i].lr, seg[i].weight_decay = a, b, 1e-2, wd
            seg[i].step_size, seg[i].bias_corr2_sqrt = 1e-2 / (1 - 0.9 ** t), math.sqrt(1 - 0.999 ** t)
            seg[i].adamw = int(adamw)
        assert L.nemo_adam_step(2, seg, p.data_ptr(), H.dev(gr).data_ptr(), m.data_ptr(), v.data_ptr(), 0.9,
                                0.999, 1e-8, H.st()) == 0
    assert rel_err(p, pt.detach()) < 1e-6


def test_pose_bwd_fused_equals_the_three_launches(L):
    """nemo_pose_bwd_fused == nemo_v2v_prep_bwd -> nemo_rot6d_bwd -> nemo_scale_neg_rowsum on the same inputs,
    with and without the optional parts (1e-6: two kernels compiled from the same expressions contract
    their FMAs differently)."""
    H = _ops()
    g = torch.Generator().manual_seed(11)
    N, LD = 37, 148
    head = H.dev(torch.randn(N + 1, LD, generator=g))
    head[:, :144] += H.dev(torch.tensor([1., 0, 0, 1, 0, 0]).repeat(24))
    dR, dAA = H.dev(torch.randn(N, 24, 9, generator=g)), H.dev(torch.randn(N, 72, generator=g))
    dAA[::3] = 0                                             # rows without an axis-angle gradient
    AA, dR2 = H.dev(0.4 * torch.randn(N, 72, generator=g)), H.dev(torch.randn(N, 24, 9, generator=g))
    dhead0 = H.dev(torch.randn(N + 1, LD, generator=g))
    scale = 0.37
    for v2v, anchored in ((True, True), (False, True), (True, False), (False, False)):
        a_dR, a_dAA, a_dh = dR.clone(), dAA.clone(), dhead0.clone()
        if v2v:
            assert L.nemo_v2v_prep_bwd(N, AA.data_ptr(), dR2.data_ptr(), scale, a_dAA.data_ptr(), a_dR.data_ptr(),
                                       H.st()) == 0
        assert L.nemo_rot6d_bwd(N, 24, head.data_ptr(), LD, 1, a_dR.data_ptr(), a_dAA.data_ptr(), a_dh.data_ptr(),
                                LD, H.st()) == 0
        if anchored:
            assert L.nemo_scale_neg_rowsum(N, 3, a_dh.data_ptr() + 4 * 144, LD,
                                           a_dh.data_ptr() + 4 * (N * LD + 144), H.st()) == 0
        b_dh = dhead0.clone()
        hmeta = torch.zeros(64, device='cuda')
        assert L.nemo_pose_bwd_fused(N, head.data_ptr(), LD, 1, dR.data_ptr(), dAA.data_ptr(), b_dh.data_ptr(), LD,
                                     AA.data_ptr() if v2v else None, dR2.data_ptr() if v2v else None, scale,
                                     b_dh.data_ptr() + 4 * 144 if anchored else None, LD, 1 if v2v else 0, hmeta.data_ptr(), H.st()) == 0
        # head_meta: the absmax of what the launch wrote (rotation columns; with dTR the translation columns incl. row N)
        cols = 147 if anchored else 144
        assert H.meta_amax(hmeta) == float(torch.cat([b_dh[:N, :144].reshape(-1), b_dh[:N + 1, 144:cols].reshape(-1)]).abs().max())
        # zero_row: the rotation columns of row N are cleared (else untouched)
        assert torch.equal(b_dh[N, :144], torch.zeros_like(b_dh[N, :144]) if v2v else dhead0[N, :144])
        assert rel_err(b_dh[:N, :144], a_dh[:N, :144]) < 1e-6, (v2v, anchored)
        assert rel_err(b_dh[N, 144:147], a_dh[N, 144:147]) < 1e-6
        assert torch.equal(b_dh[:, 147:], dhead0[:, 147:]) and torch.equal(b_dh[:N, 144:147], dhead0[:N, 144:147])
        if not anchored:
            assert torch.equal(b_dh[N, 144:147], dhead0[N, 144:147])


def test_publish_scalars_to_pinned_host_memory(L):
    """nemo_publish_scalars: device values + flag land in pinned host memory without a stream sync."""
    import time
    from nemo_cvpr2023_amd._lib import check
    H = _ops()
    src = H.dev(torch.arange(8, dtype=torch.float32) * 1.5 + 0.25)
    host = torch.zeros(16, dtype=torch.float32).pin_memory()
    flag = host.numpy().view(np.int32)[8:9]
    for rep in range(3):
        src.mul_(2.0)
        want = src.cpu().numpy()
        flag[0] = 0
        check(L.nemo_publish_scalars(src.data_ptr(), 8, host.data_ptr(), host.data_ptr() + 32, H.st()),
              'nemo_publish_scalars')
        t0 = time.monotonic()
        while flag[0] == 0:
            assert time.monotonic() - t0 < 10.0, 'flag never raised'
        assert np.array_equal(host.numpy()[:8], want)
    assert L.nemo_publish_scalars(None, 8, host.data_ptr(), host.data_ptr() + 32, H.st()) < 0
    assert L.nemo_publish_scalars(src.data_ptr(), 65, host.data_ptr(), host.data_ptr() + 32, H.st()) < 0


@pytest.mark.parametrize('ta,tb', [(0, 1), (0, 0), (1, 0)])
def test_gemm_xcd_ordered_large_launch(L, ta, tb):
    """Launches of >= 2048 whole tiles walk the tiles in an XCD-aware order (8 * ceil(tiles_m / 8) * tiles_n blocks,
    the surplus ones exit): every tile must still be computed exactly once, ragged edges included."""
    H = _ops()
    M, N, K = 4100 + 13, 2100 + 5, 72
    g = torch.Generator().manual_seed(M + ta + 2 * tb)
    pad = lambda n: (n + 3) // 4 * 4
    A = H.dev(torch.randn((K, pad(M)) if ta else (M, pad(K)), generator=g))[:, :(M if ta else K)]
    B = H.dev(torch.randn((N, pad(K)) if tb else (K, pad(N)), generator=g))[:, :(K if tb else N)]
    bias = H.dev(torch.randn(N, generator=g))
    C = torch.full((M, N), float('nan'), device='cuda')
    H.gemm(A, B, ta, tb, bias=bias, act=1, C=C, split_k=0)
    a, b = (A.double().T if ta else A.double()), (B.double().T if tb else B.double())
    assert rel_err(C, torch.relu(a @ b + bias.double())) < TOL


@pytest.mark.parametrize('n', [5, 8 * 5, 256 * 10])
def test_sqmean_instance_regulariser(L, n):
    """nemo_sqmean_fwd_bwd == (code ** 2).mean() and its gradient (nemo/neural_motion_model.py:3864-3867), added onto
    what the outputs already held; bit-identical from run to run (fixed summation order)."""
    g = torch.Generator().manual_seed(n)
    x = torch.randn(n, generator=g).cuda()
    out = torch.full((1,), 0.25, device='cuda')
    grad = torch.ones(n, device='cuda')
    st = torch.cuda.current_stream().cuda_stream
    assert L.nemo_sqmean_fwd_bwd(n, x.data_ptr(), out.data_ptr(), grad.data_ptr(), 0.5, st) == 0
    torch.testing.assert_close(out.cpu(), 0.25 + (x.cpu().double() ** 2).mean().float().reshape(1), rtol=2e-6, atol=0)
    torch.testing.assert_close(grad.cpu(), 1.0 + 0.5 * x.cpu(), rtol=1e-6, atol=0)
    out2 = torch.full((1,), 0.25, device='cuda')
    assert L.nemo_sqmean_fwd_bwd(n, x.data_ptr(), out2.data_ptr(), None, 0.0, st) == 0
    assert torch.equal(out2, out)
    assert L.nemo_sqmean_fwd_bwd(0, x.data_ptr(), out.data_ptr(), None, 0.0, st) != 0


@pytest.mark.parametrize('ta,tb,M,N,K,pad', [
    (0, 1, 300, 1000, 1000, 4),      # forward, 32 x 64 tiles on 8 waves
    (0, 1, 301, 147, 1000, 4),       # forward head, 32 x 32 tiles on 4 waves
    (0, 1, 300, 1000, 105, 0),       # weights with in_features = 105: rows not 16-byte aligned
    (0, 1, 30, 512, 63, 0),          # C1-sized batch, K not a multiple of 8, unaligned
    (0, 0, 301, 1000, 147, 4),       # input gradient
    (0, 0, 1200, 1000, 1000, 4),     # input gradient of a 4-instance shard
    (1, 0, 1000, 1000, 333, 4),      # parameter gradient, 64 x 64 tiles on 8 waves (128 KiB reduction buffer)
    (1, 0, 1000, 1000, 2401, 4),
    (1, 0, 147, 1000, 301, 4),
    (1, 0, 1000, 105, 300, 1),       # dW of the first layer (odd ld on both sides)
    (1, 1, 77, 207, 515, 4),
    (1, 1, 300, 207, 20670, 4),      # blend-shape adjoint of a one-instance shard: K also cut across blocks (16 slices)
    (1, 1, 600, 207, 20670, 4),      # ... of a two-instance shard: ONE 32 x 224 column tile, 13 K slices (round 3)
    (1, 1, 450, 207, 4100, 0),       # same kernel, ragged rows, K not a multiple of 8 x slices
])
def test_gemm_skinny_paths(L, ta, tb, M, N, K, pad):
    """Problems the auto plan sends to the intra-block K-split kernel (csrc/gemm_skinny.h): product, fused bias + ReLU,
    masked C += (the MLP backward's epilogue), bit-identical repeat, nothing written outside C's logical columns."""
    H = _ops()
    g = torch.Generator().manual_seed(M + 3 * N + 5 * K + ta + 2 * tb)
    lda = (M if ta else K) + pad
    ldb = (K if tb else N) + pad
    Af = torch.randn((K if ta else M), lda, generator=g)
    Bf = torch.randn((N if tb else K), ldb, generator=g)
    Af[:, (M if ta else K):] = float('nan')                 # pads must never reach a result
    Bf[:, (K if tb else N):] = float('nan')
    A, B = Af[:, :(M if ta else K)], Bf[:, :(K if tb else N)]
    bias, mask = torch.randn(N, generator=g), torch.randn(M, N, generator=g)
    ref = (A.T if ta else A).double() @ (B.T if tb else B).double()
    dA, dB = H.dev(Af)[:, :A.shape[1]], H.dev(Bf)[:, :B.shape[1]]
    C1 = H.gemm(dA, dB, ta, tb, split_k=0, bias=H.dev(bias), act=1)
    assert rel_err(C1, torch.relu(ref + bias.double())) < TOL
    C2 = H.gemm(dA, dB, ta, tb, split_k=0, bias=H.dev(bias), act=1)
    assert torch.equal(C1, C2)
    C0 = torch.randn(M, N + 3, generator=g)
    Cw = H.dev(C0).clone()
    H.gemm(dA, dB, ta, tb, split_k=0, mask=H.dev(mask), mask_mode=1, out_mode=1, C=Cw[:, :N])
    assert rel_err(Cw[:, :N], C0[:, :N].double() + torch.where(mask > 0, ref, torch.zeros_like(ref))) < TOL
    assert torch.equal(Cw[:, N:].cpu(), C0[:, N:])
    assert int(H.gemm_ws()[:4096].view(torch.int32).abs().sum()) == 0        # tickets of the sliced launches back at zero


@pytest.mark.parametrize('rows', [2401, 301, 37])
@pytest.mark.parametrize('bf16', [False, True])
def test_gemm_grouped_parameter_gradients(L, rows, bf16):
    """nemo_gemm_grouped_f32 / _bf16: the four parameter-gradient products dW = dY^T X of the MotionNet backward (heads 147,
    two hidden layers, first layer with its 108-float row stride) in ONE launch -- `C +=` on the two big ones and a plain
    store on the others, the small problems cut along K at 2401 rows -- against float64; a repeat is bit-identical; the
    tickets are back at zero; a group of mixed layouts falls back to single launches with the same results."""
    from nemo_cvpr2023_amd._lib import GemmProblem, check
    H = _ops()
    g = torch.Generator().manual_seed(rows + bf16)
    h, din, ldx = 1000, 105, 108
    dHEAD, H3 = H.dev(torch.randn(rows, 148, generator=g)), H.dev(torch.randn(rows, h, generator=g))
    dH, H2 = H.dev(torch.randn(rows, h, generator=g)), H.dev(torch.randn(rows, h, generator=g))
    dHb, H1 = H.dev(torch.randn(rows, h, generator=g)), H.dev(torch.randn(rows, h, generator=g))
    dHc, X = H.dev(torch.randn(rows, h, generator=g)), H.dev(torch.randn(rows, ldx, generator=g))
    probs = [(147, h, dHEAD, 148, H3, h, 1), (h, h, dH, h, H2, h, 1), (h, h, dHb, h, H1, h, 0), (h, din, dHc, h, X, ldx, 1)]
    fn = L.nemo_gemm_grouped_bf16 if bf16 else L.nemo_gemm_grouped_f32
    tol = 1e-2 if bf16 else TOL
    ws = H.gemm_ws()

    def run(problems):
        C0 = [H.dev(torch.randn(M, N, generator=torch.Generator().manual_seed(i))) for i, (M, N, *_) in enumerate(problems)]
        Cs = [c.clone() for c in C0]
        arr = (GemmProblem * len(problems))()
        for i, (M, N, A, lda, B, ldb, om) in enumerate(problems):
            q = arr[i]
            q.transA, q.transB, q.M, q.N, q.K = 1, 0, M, N, rows
            q.A, q.lda, q.B, q.ldb, q.C, q.ldc, q.alpha, q.out_mode = A.data_ptr(), lda, B.data_ptr(), ldb, Cs[i].data_ptr(), N, 1.0, om
        check(fn(len(problems), arr, ws.data_ptr(), ws.numel() * 4, H.st()), 'grouped')
        return C0, Cs
    C0, Cs = run(probs)
    for (M, N, A, lda, B, ldb, om), c0, c in zip(probs, C0, Cs):
        ref = A[:, :M].double().T @ B[:, :N].double() + (c0.double() if om else 0)
        assert rel_err(c, ref) < tol, (M, N, om)
    _, Cs2 = run(probs)
    assert all(torch.equal(a, b) for a, b in zip(Cs, Cs2))
    assert int(ws[:4096].view(torch.int32).abs().sum()) == 0
    # two problems only; and a single one (falls back to nemo_gemm_f32)
    for sub in (probs[:2], probs[1:2]):
        C0, Cs = run(sub)
        for (M, N, A, lda, B, ldb, om), c0, c in zip(sub, C0, Cs):
            assert rel_err(c, A[:, :M].double().T @ B[:, :N].double() + (c0.double() if om else 0)) < tol


def test_gemm_auto_plan_random_shapes(L):
    """split_k = 0 over random problems on both sides of every kernel-selection threshold (skinny 32x32 / 32x64 /
    64x64 / sliced, LDS-DMA tiles, first-generation kernel for unaligned rows): product, bias + ReLU, determinism."""
    H = _ops()
    rng = np.random.default_rng(20230)
    dims_m = [1, 30, 31, 300, 301, 512, 600, 1024, 1100, 1537, 2400]
    dims_n = [3, 64, 105, 147, 207, 512, 1000]
    dims_k = [5, 63, 105, 207, 300, 512, 1000, 2401]
    for case in range(40):
        ta, tb = int(rng.integers(0, 2)), int(rng.integers(0, 2))
        M, N, K = int(rng.choice(dims_m)), int(rng.choice(dims_n)), int(rng.choice(dims_k))
        if M * N * K > 1.3e9:
            K = 300
        pad_a, pad_b = int(rng.choice([0, 1, 4])), int(rng.choice([0, 3, 4]))
        g = torch.Generator().manual_seed(1000 + case)
        Af = torch.randn((K if ta else M), (M if ta else K) + pad_a, generator=g)
        Bf = torch.randn((N if tb else K), (K if tb else N) + pad_b, generator=g)
        A, B = Af[:, :(M if ta else K)], Bf[:, :(K if tb else N)]
        bias = torch.randn(N, generator=g)
        ref = torch.relu((A.T if ta else A).double() @ (B.T if tb else B).double() + bias.double())
        dA, dB = H.dev(Af)[:, :A.shape[1]], H.dev(Bf)[:, :B.shape[1]]
        C1 = H.gemm(dA, dB, ta, tb, split_k=0, bias=H.dev(bias), act=1)
        assert rel_err(C1, ref) < TOL, (case, ta, tb, M, N, K, pad_a, pad_b)
        C2 = H.gemm(dA, dB, ta, tb, split_k=0, bias=H.dev(bias), act=1)
        assert torch.equal(C1, C2), (case, ta, tb, M, N, K)
    assert int(H.gemm_ws()[:4096].view(torch.int32).abs().sum()) == 0


def _mesh_term_f64(assets, PF, A, N):
    """The fused mesh term evaluated in float64 FROM THE KERNEL'S fp32 INPUTS (pose features PF (2N, 208), transforms A
    (2N, 24, 12), rows [orig | reconstruction]): loss, d loss / d vp_orig (N, NV, 3), d loss / d A_orig (N, 24, 12) and the
    per-element differences d = v_rec - v_orig (lbs.py:229-252, neural_motion_model.py:2787-2793)."""
    dev = PF.device
    P = assets['posedirs'].to(dev).double()                          # (207, 3 NV)
    W = assets['lbs_weights'].to(dev).double()                       # (NV, 24)
    vt = assets['v_template'].to(dev).double()                       # (NV, 3): betas are zero
    NV = vt.shape[0]
    vp = vt.unsqueeze(0) + (PF[:, :207].double() @ P).reshape(2 * N, NV, 3)
    T = torch.einsum('vj,nje->nve', W, A.double().reshape(2 * N, 24, 12)).reshape(2 * N, NV, 3, 4)
    vert = (T[..., :3] @ vp.unsqueeze(-1)).squeeze(-1) + T[..., 3]
    d = vert[N:] - vert[:N]
    g = -torch.sign(d)                                                # d |v_rec - v_orig| / d v_orig
    dvp = (T[:N, :, :, :3] * g.unsqueeze(-1)).sum(2)                  # T^T g
    dT = g.unsqueeze(-1) * torch.cat([vp[:N], torch.ones_like(vp[:N, :, :1])], -1).unsqueeze(2)     # (N, NV, 3, 4)
    dA = torch.einsum('vj,nve->nje', W, dT.reshape(N, NV, 12))
    return d.abs().sum(), dvp, dA, d


@pytest.mark.parametrize('skin_nnz', [4, 24])
def test_v2v_fused_split_is_fp32_equivalent(L, skin_nnz):
    """nemo_v2v_fused_split (pose blend on the bf16 pipe, three bf16 pieces per operand) against a float64 evaluation of
    the same fp32 inputs: its error must not exceed 1.5 x the fp32-MFMA kernel's (criterion (a)) -- it
    is not narrower arithmetic than the reference's.  Blend shapes scaled x 30 so that the pose offsets are a third of
    the template (with the synthetic model's 1e-3 offsets a blend of ANY precision would hide behind the template's
    rounding; x 100, the scaling of round 5, puts the model's worst-case vertex bound outside the fp16 form's range guard --
    nemo_ctx_split_ok -- and the three-piece form would run: that case is test_v2v_fused_split_regimes_and_range_guard's)."""
    H = _ops()
    num_verts, N = 6890, 40
    assets = syn.make_smpl_assets(num_verts, seed=1, skin_nnz=skin_nnz)
    assets['posedirs'] = assets['posedirs'] * 30.0
    from nemo_cvpr2023_amd.engine import SmplContext
    jm = [int(x) for x in assets['joint_map']]
    ctx = SmplContext(assets, [jm[i] for i in [38] + list(range(1, 25))], 'cuda:0')
    assert ctx.split_ok, ctx.vp_bound
    gen = torch.Generator().manual_seed(3)
    R2 = H.dev(_rand_rot(gen, 2 * N * 24, small=True).reshape(2 * N, 24, 9))
    Z = lambda *s: torch.zeros(*s, device='cuda')
    A, Jp, PF = Z(2 * N, 24, 12), Z(2 * N, 24, 3), Z(2 * N, 208)
    assert L.nemo_fk_fwd(ctx.handle, 2 * N, R2.data_ptr(), A.data_ptr(), Jp.data_ptr(), PF.data_ptr(), 208, H.st()) == 0
    torch.cuda.synchronize()
    l_ref, dvp_ref, dA_ref, d = _mesh_term_f64(assets, PF, A, N)
    # samples with an element within rounding of a tie may take the other sign in either kernel: excluded from the gradients
    clean = (d.abs().reshape(N, -1).min(1).values > 3e-6)
    assert int(clean.sum()) >= N // 2
    ldn = (N + 15) // 16 * 16
    ws = torch.zeros(int(L.nemo_v2v_fused_ws_bytes(ctx.handle, N)) // 4 + 1, device='cuda')
    err = {}
    for name, fn in (('f32', L.nemo_v2v_fused), ('split', L.nemo_v2v_fused_split)):
        loss, dVPt, dA = Z(1), Z(3 * ctx.NVp, ldn), Z(N, 24, 12)
        assert fn(ctx.handle, N, PF.data_ptr(), 208, A.data_ptr(), loss.data_ptr(), dVPt.data_ptr(), ldn, dA.data_ptr(),
                  ws.data_ptr(), ws.numel() * 4, H.st()) == 0
        torch.cuda.synchronize()
        dvp = dVPt[:3 * num_verts, :N].t().reshape(N, num_verts, 3).double()
        e_dA = (dA.double() - dA_ref)[clean]
        err[name] = dict(loss=abs(float(loss) - float(l_ref)) / float(l_ref),
                         dA_max=float(e_dA.abs().max() / dA_ref.abs().max()),
                         dA_rms=float(e_dA.pow(2).mean().sqrt() / dA_ref.pow(2).mean().sqrt()),
                         dvp_max=float((dvp - dvp_ref)[clean].abs().max() / dvp_ref.abs().max()))
    print('mesh term error against float64:', err)
    f, s3 = err['f32'], err['split']
    assert s3['dA_rms'] <= 1.5 * f['dA_rms'] + 1e-9, err
    assert s3['dA_max'] <= 1.5 * f['dA_max'] + 1e-9, err
    assert s3['dvp_max'] <= 1.5 * f['dvp_max'] + 1e-9, err
    assert s3['loss'] <= 1.5 * f['loss'] + 2e-7, err              # (one fp32 scalar: both sit at its rounding)
    assert s3['dA_max'] < 1e-5 and s3['loss'] < 1e-5


@pytest.mark.parametrize('M,N,K,ta,tb', [(2401, 1000, 1000, 0, 0), (2401, 1000, 147, 0, 0), (700, 130, 96, 0, 1), (333, 77, 105, 0, 0),
                                          (8200, 1000, 1000, 0, 0)])
def test_gemm_f32_with_per_band_column_sums(L, M, N, K, ta, tb):
    """nemo_gemm_f32_colsum: the product of nemo_gemm_f32 (bit-identical C) plus the column sums of the masked result per 32-row
    band -- from the LDS-DMA kernel's epilogue when the plan is that kernel, from C otherwise (unaligned K = 105 rows)."""
    H = _ops()
    gen = torch.Generator().manual_seed(M + N)
    A = H.dev(torch.randn(M, K, generator=gen))
    B = H.dev(torch.randn(*((N, K) if tb else (K, N)), generator=gen))
    mask = H.dev(torch.randn(M, N, generator=gen))
    ws = torch.zeros((16384 + 65536 + (64 << 20)) // 4, device='cuda')
    C1, C2 = torch.zeros(M, N, device='cuda'), torch.zeros(M, N, device='cuda')
    R = int(L.nemo_gemm_colsum_rows(M))
    cs = torch.full((R, N + 3), 7.0, device='cuda')
    assert L.nemo_gemm_f32(ta, tb, M, N, K, A.data_ptr(), K, B.data_ptr(), B.stride(0), C1.data_ptr(), N, None, 0,
                           mask.data_ptr(), N, 1, 1.0, 0, 0, ws.data_ptr(), ws.numel() * 4, H.st()) == 0
    assert L.nemo_gemm_f32_colsum(ta, tb, M, N, K, A.data_ptr(), K, B.data_ptr(), B.stride(0), C2.data_ptr(), N, None, 0,
                                  mask.data_ptr(), N, 1, 1.0, cs.data_ptr(), cs.stride(0), ws.data_ptr(), ws.numel() * 4, H.st()) == 0
    torch.cuda.synchronize()
    ref = (A.double() @ (B.double().t() if tb else B.double())) * (mask > 0)
    assert rel_err(C2, ref) < 2e-5
    if M > 1536 or N < 100:                      # (small M with aligned operands: nemo_gemm_f32 alone takes the skinny kernel)
        assert torch.equal(C1, C2)
    pad = torch.zeros(R * 32 - M, N, device='cuda', dtype=torch.float64)
    bands = torch.cat([C2.double(), pad]).reshape(R, 32, N).sum(1)
    assert rel_err(cs[:, :N], bands) < 1e-5
    assert float((cs[:, N:] - 7.0).abs().max()) == 0.0                           # columns beyond N untouched
    assert rel_err(cs[:, :N].double().sum(0), ref.sum(0)) < 1e-4


@pytest.mark.parametrize('M,K', [(2400, 20670), (300, 20670), (513, 4098), (8192, 20670), (257, 1030)])
def test_blend_shape_adjoint_in_split_precision(L, M, K):
    """nemo_gemm_f16x2mem_adj: C += alpha (A0 B0^T + A0 B1^T + A1 B0^T) over two fp16 piece planes per operand (the 64 x 208 tile, K
    slices dealt over the three plane pairs) against float64 -- and no worse than 1.5 x the fp32 GEMM on the same operands."""
    H = _ops()
    N = 207
    gen = torch.Generator().manual_seed(M + K)
    A = torch.randn(M, K, generator=gen) * torch.rand(M, 1, generator=gen)          # d vp-like: O(1), row scales differ
    B = 1e-2 * torch.randn(N, K, generator=gen)
    sa, sb = 4096.0, 2.0 ** (14 - math.frexp(float(B.abs().max()))[1])

    def planes(x, s, ld):
        y = x * s
        h0 = y.half()
        h1 = (y - h0.float()).half()
        out = torch.zeros(2, x.shape[0], ld, dtype=torch.int16)
        out[0, :, :x.shape[1]] = h0.view(torch.int16)
        out[1, :, :x.shape[1]] = h1.view(torch.int16)
        return out.cuda()
    ld = (K + 15) // 8 * 8
    Ah, Bh = planes(A, sa, ld), planes(B, sb, ld)
    ws = torch.zeros((16384 + 65536 + (96 << 20)) // 4, device='cuda')
    C0 = torch.randn(M, 208, generator=gen).cuda()
    C = C0.clone()
    K2 = (K + 1) // 2 * 2
    assert L.nemo_gemm_f16x2mem_adj(M, N, K2, Ah.data_ptr(), ld, Ah.stride(0), Bh.data_ptr(), ld, Bh.stride(0), C.data_ptr(), 208,
                                    1.0 / (sa * sb), 1, ws.data_ptr(), ws.numel() * 4, H.st()) == 0
    torch.cuda.synchronize()
    assert float(ws[:4096].abs().max()) == 0.0                                    # tickets back at zero
    ref = A.double() @ B.double().t()
    got = (C[:, :N] - C0[:, :N]).double().cpu()
    e_split = float((got - ref).abs().max() / ref.abs().max())
    # the fp32 product of the same operands through the library (transposed A, as the fp32 build runs it)
    At = H.dev(A.t().contiguous())
    Bd = H.dev(B)
    C32 = torch.zeros(M, 208, device='cuda')
    assert L.nemo_gemm_f32(1, 1, M, N, K, At.data_ptr(), M, Bd.data_ptr(), K, C32.data_ptr(), 208, None, 0, None, 0, 0, 1.0, 0, 0,
                           ws.data_ptr(), ws.numel() * 4, H.st()) == 0
    torch.cuda.synchronize()
    e32 = float((C32[:, :N].double().cpu() - ref).abs().max() / ref.abs().max())
    print('blend-shape adjoint, max error / max |result| against float64: split', e_split, 'fp32 kernel', e32)
    assert e_split <= 1.5 * e32 + 1e-7 and e_split < 3e-6
    assert torch.equal(C[:, N:], C0[:, N:])                                       # column 207 untouched


@pytest.mark.parametrize('regime', ['model_x2.5', 'model_x10', 'model_x100', 'pose_1e-5', 'pose_1e-7', 'posedirs_6_decades', 'vp_at_the_edge'])
def test_v2v_fused_split_regimes_and_range_guard(L, regime):
    """nemo_v2v_fused_split over the magnitude regimes its fp16 pieces could be sensitive to, each against a
    float64 evaluation of the same fp32 inputs and against the fp32-MFMA kernel's error (<= 1.5 x, as in the test above):
      model_x2.5 / x10 / x100  the body model in other units: v_template and posedirs x 2.5 (inside the fp16 form's range: 2^12 x the bound
                        on |vp| < 2^15.9), x 10 and x 100 (outside: nemo_ctx_split_ok must say so and the three-bf16-piece form must run
                        -- no inf / NaN, same error bar);
      pose_1e-5 / 1e-7  pose features of the "orig" body at the published initialisation's size (rotations within 1e-5 / 1e-7 rad of
                        identity: the low fp16 piece of a pose feature is a subnormal or zero) against a reconstruction body at 0.3 rad
                        -- what the step sees at its first iterations (the VPoser reconstruction is NOT near identity);
      posedirs_6_decades  blend shapes whose entries spread log-uniformly over six decades (real SMPL spans > 4);
      vp_at_the_edge    a model scaled so that the bound sits just inside the guard (2^12 bound = 0.97 x 61 000)."""
    H = _ops()
    num_verts, N = 6890, 40
    assets = syn.make_smpl_assets(num_verts, seed=1, skin_nnz=4)
    assets['posedirs'] = assets['posedirs'] * 10.0
    small = 0.3
    gen = torch.Generator().manual_seed(7)
    if regime.startswith('model_x'):
        k = float(regime.split('x')[1])
        assets['v_template'] = assets['v_template'] * k
        assets['posedirs'] = assets['posedirs'] * k
    elif regime.startswith('pose_'):
        small = float(regime.split('_')[1])
    elif regime == 'posedirs_6_decades':
        P = assets['posedirs']
        assets['posedirs'] = P * torch.pow(10.0, -6.0 * torch.rand(P.shape, generator=gen))
    from nemo_cvpr2023_amd.engine import SmplContext
    jm = [int(x) for x in assets['joint_map']]
    oj = [jm[i] for i in [38] + list(range(1, 25))]
    ctx = SmplContext(assets, oj, 'cuda:0')
    if regime == 'vp_at_the_edge':
        k = 0.97 * 61000.0 / 4096.0 / ctx.vp_bound
        assets['v_template'] = assets['v_template'] * k
        assets['posedirs'] = assets['posedirs'] * k
        ctx = SmplContext(assets, oj, 'cuda:0')
        assert ctx.split_ok and ctx.vp_bound * 4096 > 0.9 * 61000
    assert ctx.split_ok == (regime not in ('model_x10', 'model_x100')), (regime, ctx.vp_bound)
    from oracle import ops
    th = torch.randn(2 * N * 24, 3, generator=gen, dtype=torch.float64) * small
    if regime.startswith('pose_'):
        th[N * 24:] *= 0.3 / small
    R2 = H.dev(ops.batch_rodrigues(th).float().reshape(2 * N, 24, 9))
    Z = lambda *s: torch.zeros(*s, device='cuda')
    A, Jp, PF = Z(2 * N, 24, 12), Z(2 * N, 24, 3), Z(2 * N, 208)
    assert L.nemo_fk_fwd(ctx.handle, 2 * N, R2.data_ptr(), A.data_ptr(), Jp.data_ptr(), PF.data_ptr(), 208, H.st()) == 0
    torch.cuda.synchronize()
    l_ref, dvp_ref, dA_ref, d = _mesh_term_f64(assets, PF, A, N)
    vmax = float(assets['v_template'].abs().max())
    clean = (d.abs().reshape(N, -1).min(1).values > 3e-6 * max(vmax, 1.0))
    ldn = (N + 15) // 16 * 16
    ws = torch.zeros(int(L.nemo_v2v_fused_ws_bytes(ctx.handle, N)) // 4 + 1, device='cuda')
    err = {}
    for name, fn in (('f32', L.nemo_v2v_fused), ('split', L.nemo_v2v_fused_split)):
        loss, dVPt, dA = Z(1), Z(3 * ctx.NVp, ldn), Z(N, 24, 12)
        assert fn(ctx.handle, N, PF.data_ptr(), 208, A.data_ptr(), loss.data_ptr(), dVPt.data_ptr(), ldn, dA.data_ptr(),
                  ws.data_ptr(), ws.numel() * 4, H.st()) == 0
        torch.cuda.synchronize()
        assert torch.isfinite(loss).all() and torch.isfinite(dA).all() and torch.isfinite(dVPt).all(), (regime, name)
        dvp = dVPt[:3 * num_verts, :N].t().reshape(N, num_verts, 3).double()
        e_dA = (dA.double() - dA_ref)[clean]
        err[name] = dict(loss=abs(float(loss) - float(l_ref)) / max(float(l_ref), 1e-30),
                         dA_max=float(e_dA.abs().max() / dA_ref.abs().max()),
                         dA_rms=float(e_dA.pow(2).mean().sqrt() / dA_ref.pow(2).mean().sqrt()),
                         dvp_max=float((dvp - dvp_ref)[clean].abs().max() / dvp_ref.abs().max()))
    print(regime, 'mesh term error against float64:', err)
    f, s = err['f32'], err['split']
    assert int(clean.sum()) >= N // 2
    assert s['dA_rms'] <= 1.5 * f['dA_rms'] + 1e-9, err
    assert s['dA_max'] <= 1.5 * f['dA_max'] + 1e-9, err
    assert s['dvp_max'] <= 1.5 * f['dvp_max'] + 1e-9, err
    assert s['loss'] <= 1.5 * f['loss'] + 2e-7, err
    # fp16 piece planes of d vp are refused outside the range (the engine keeps d vp in fp32 then)
    if not ctx.split_ok:
        dh = torch.zeros(2, ldn, ctx.ldP, dtype=torch.int16, device='cuda')
        assert L.nemo_v2v_fused_splitmem(ctx.handle, N, PF.data_ptr(), 208, A.data_ptr(), Z(1).data_ptr(), dh.data_ptr(), dh.stride(1),
                                         dh.stride(0), Z(N, 24, 12).data_ptr(), ws.data_ptr(), ws.numel() * 4, H.st()) < 0


def test_smooth_kernel_against_the_reference_method(L):
    """nemo_smooth_fwd_bwd against the fixture recorded from the reference's own `FittingLoss.joints3d_smooth_loss`
    (humor/humor/fitting/fitting_loss.py:366-370): value and gradient, three shapes incl. T = 2."""
    from conftest import load_golden
    H = _ops()
    g = load_golden('fn_joints3d_smooth_loss')
    for tag in 'abc':
        j = torch.tensor(g[f'{tag}_joints'])
        B, T_, J, _ = j.shape
        dj, out, dd = H.dev(j.reshape(B * T_, J, 3)), torch.zeros(1, device='cuda'), torch.zeros(B * T_, J, 3, device='cuda')
        for w in (1.0, 2.5):
            out.zero_()
            assert L.nemo_smooth_fwd_bwd(B, T_, J, dj.data_ptr(), w, out.data_ptr(), dd.data_ptr(), H.st()) == 0
            assert rel_err(out.reshape(()), g[f'{tag}_loss']) < 1e-5
            assert rel_err(dd.reshape(B, T_, J, 3), w * torch.tensor(g[f'{tag}_grad'])) < 1e-5


def test_blend_shape_adjoint_split_precision_pairs(L):
    """The blend-shape adjoint dPF = d vp P^T (lbs.py:229-233 backward) of fp32 builds, three ways on the same inputs:
      (a) nemo_v2v_fused_split -> fp32 d vp^T -> nemo_gemm_f32;
      (b) round 5: nemo_v2v_fused_splitmem (two fp16 piece planes) -> nemo_gemm_f16x2mem_adj;
      (c) round 6, the engine's default: nemo_v2v_fused_splitxp (one xp matrix) -> nemo_gemm_xp against the blend shapes' xp copy.
    (b) and (c) must hold the SAME pieces of d vp, and their dPF must be as close to a float64 product as (a)'s."""
    H = _ops()
    num_verts, N = 6890, 320
    assets = syn.make_smpl_assets(num_verts, seed=1, skin_nnz=4)
    assets['posedirs'] = assets['posedirs'] * 30.0
    from nemo_cvpr2023_amd.engine import SmplContext
    jm = [int(x) for x in assets['joint_map']]
    ctx = SmplContext(assets, [jm[i] for i in [38] + list(range(1, 25))], 'cuda:0')
    gen = torch.Generator().manual_seed(5)
    R2 = H.dev(_rand_rot(gen, 2 * N * 24, small=True).reshape(2 * N, 24, 9))
    Z = lambda *s: torch.zeros(*s, device='cuda')
    A, Jp, PF = Z(2 * N, 24, 12), Z(2 * N, 24, 3), Z(2 * N, 208)
    assert L.nemo_fk_fwd(ctx.handle, 2 * N, R2.data_ptr(), A.data_ptr(), Jp.data_ptr(), PF.data_ptr(), 208, H.st()) == 0
    _, dvp_ref, _, _ = _mesh_term_f64(assets, PF, A, N)
    P64 = assets['posedirs'].to('cuda').double()                                  # (207, 3 NV)
    dPF_ref = dvp_ref.reshape(N, -1) @ P64.t()
    ldn, NV3 = (N + 15) // 16 * 16, 3 * num_verts
    ws = torch.zeros(int(L.nemo_v2v_fused_ws_bytes(ctx.handle, N)) // 4 + 1, device='cuda')
    gws = H.gemm_ws()
    # (a)
    loss, dVPt, dA = Z(1), Z(3 * ctx.NVp, ldn), Z(N, 24, 12)
    assert L.nemo_v2v_fused_split(ctx.handle, N, PF.data_ptr(), 208, A.data_ptr(), loss.data_ptr(), dVPt.data_ptr(), ldn, dA.data_ptr(),
                                  ws.data_ptr(), ws.numel() * 4, H.st()) == 0
    Pd = torch.zeros(207, ctx.ldP, device='cuda')
    Pd[:, :NV3] = assets['posedirs'].to('cuda')
    dPF_a = H.gemm(dVPt[:NV3], Pd[:, :NV3], 1, 1)
    # (b)
    dh = torch.zeros(2, ldn, ctx.ldP, dtype=torch.int16, device='cuda')
    assert L.nemo_v2v_fused_splitmem(ctx.handle, N, PF.data_ptr(), 208, A.data_ptr(), Z(1).data_ptr(), dh.data_ptr(), dh.stride(1), dh.stride(0),
                                     Z(N, 24, 12).data_ptr(), ws.data_ptr(), ws.numel() * 4, H.st()) == 0
    x = Pd * 1.0
    pmax = float(x.abs().max())
    import math
    ps = 2.0 ** (14 - math.frexp(pmax)[1])
    h0 = (x * ps).half()
    h1 = (x * ps - h0.float()).half()
    Ph = torch.stack([h0.view(torch.int16), h1.view(torch.int16)])
    dPF_b = Z(N, 208)
    assert L.nemo_gemm_f16x2mem_adj(N, 207, (NV3 + 1) // 2 * 2, dh.data_ptr(), dh.stride(1), dh.stride(0), Ph.data_ptr(), Ph.stride(1), Ph.stride(0),
                                    dPF_b.data_ptr(), 208, 1.0 / (4096.0 * ps), 0, gws.data_ptr(), gws.numel() * 4, H.st()) == 0
    # (c)
    dx = torch.zeros(ldn, H.xp_ld(2, 3 * ctx.NVp), dtype=torch.int16, device='cuda')
    assert L.nemo_v2v_fused_splitxp(ctx.handle, N, PF.data_ptr(), 208, A.data_ptr(), Z(1).data_ptr(), dx.data_ptr(), dx.stride(0),
                                    Z(N, 24, 12).data_ptr(), ws.data_ptr(), ws.numel() * 4, H.st()) == 0
    mP = H.absmax_meta(Pd[:, :NV3])[0]
    Px, _ = H.cast_xp(2, Pd[:, :NV3], meta=mP)
    mA = torch.zeros(64, device='cuda')
    mA[0] = 4096.0
    dPF_c = Z(N, 208)
    H.gemm_xp(2, dx, Px, N, 207, NV3, C=dPF_c, metaA=mA, metaB=mP)
    torch.cuda.synchronize()
    # the two hand-overs hold the same pieces
    planes = dh[:, :N, :NV3].view(torch.float16).double().sum(0)
    assert torch.equal(H.xp_decode(2, dx[:N], NV3), planes)
    err = lambda c: float((c[:, :207].double() - dPF_ref).abs().max() / dPF_ref.abs().max())
    ea, eb, ec = err(dPF_a), err(dPF_b), err(dPF_c)
    print('blend-shape adjoint error against float64: fp32', ea, ' f16 planes', eb, ' xp', ec)
    assert eb <= 1.5 * ea + 1e-9 and ec <= 1.5 * ea + 1e-9
    # the xp route twice: same bits (ordered slab combine)
    dPF_c2 = Z(N, 208)
    H.gemm_xp(2, dx, Px, N, 207, NV3, C=dPF_c2, metaA=mA, metaB=mP)
    assert torch.equal(dPF_c, dPF_c2)
