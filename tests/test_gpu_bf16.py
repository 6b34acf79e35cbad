"""BASELINE configs[2]: the bf16 variant of the dense contractions (args.gemm_dtype = 'bf16') -- MotionNet / VPoser
linear layers forward and backward, the pose blend of the mesh term and its adjoint on the bf16 matrix cores, fp32
accumulation, fp32 master weights.  Not covered by the 1e-4 fp32 parity gate; the tolerances below are the bf16 ones:

* kernel level, exact semantics: ``nemo_gemm_bf16`` must equal an fp32-accumulated product of the operands ROUNDED TO
  bf16 (round-to-nearest-even) to 2e-5 -- i.e. the only difference to the fp32 kernel is the operand rounding;
* against the unrounded float64 product: 1e-2 of the result's scale (8 mantissa bits, K up to 1000);
* mesh term: bf16 blend vs the fp32 kernel 2e-3 on the loss, 2e-2 on the gradients;
* whole step at C3 (40 x 300, N = 12 000): losses within 5e-3 of the fp32 path, parameter gradients with cosine > 0.995 to the fp32
  ones (entries within 0.15 of the gradient's scale after three bf16 layers of backward), and the
  fit still descends."""
import numpy as np
import os
import pytest
import torch

from conftest import rel_err
from nemo_cvpr2023_amd import synthetic as syn

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


@pytest.fixture(scope='module')
def L():
    from nemo_cvpr2023_amd import _lib
    return _lib.load()


def _gemm(L, fn, A, B, ta, tb, **kw):
    import hipops as H
    from nemo_cvpr2023_amd._lib import check, dptr
    M = A.shape[1] if ta else A.shape[0]
    K = A.shape[0] if ta else A.shape[1]
    N = B.shape[0] if tb else B.shape[1]
    C = kw.pop('C', None)
    if C is None:
        C = torch.zeros(M, N, device=DEV)
    bias, act = kw.get('bias'), kw.get('act', 0)
    check(fn(ta, tb, M, N, K, dptr(A), A.stride(0), dptr(B), B.stride(0), dptr(C), C.stride(0), dptr(bias), act, None, 0, 0,
             1.0, kw.get('out_mode', 0), kw.get('split_k', 0), dptr(H.gemm_ws()), H.gemm_ws().numel() * 4, H.st()), 'gemm')
    return C


@pytest.mark.parametrize('ta,tb', [(0, 1), (0, 0), (1, 0), (1, 1)])
@pytest.mark.parametrize('M,N,K', [(301, 1000, 1000), (2401, 147, 1000), (130, 70, 100), (300, 207, 2070)])
def test_gemm_bf16_is_the_fp32_kernel_on_rounded_operands(L, ta, tb, M, N, K):
    g = torch.Generator().manual_seed(M + 3 * N + 7 * K + ta + 2 * tb)
    pad = lambda n: (n + 3) // 4 * 4
    A = torch.randn((K, pad(M)) if ta else (M, pad(K)), generator=g).to(DEV)[:, :(M if ta else K)]
    B = torch.randn((N, pad(K)) if tb else (K, pad(N)), generator=g).to(DEV)[:, :(K if tb else N)]
    bias = torch.randn(N, generator=g).to(DEV)
    C = _gemm(L, L.nemo_gemm_bf16, A, B, ta, tb, bias=bias, act=1)
    rnd = lambda x: x.to(torch.bfloat16).to(torch.float64)          # torch rounds to nearest even, like the kernel
    a, b = (rnd(A).T if ta else rnd(A)), (rnd(B).T if tb else rnd(B))
    ref_rounded = torch.relu(a @ b + bias.double())
    assert rel_err(C, ref_rounded) < 2e-5
    a, b = (A.double().T if ta else A.double()), (B.double().T if tb else B.double())
    assert rel_err(C, torch.relu(a @ b + bias.double())) < 1e-2
    # split-K in the launch, accumulate mode
    C2 = torch.ones(M, N, device=DEV)
    _gemm(L, L.nemo_gemm_bf16, A, B, ta, tb, out_mode=1, split_k=3 if K >= 192 else 1, C=C2)
    assert rel_err(C2, 1.0 + (rnd(A).T if ta else rnd(A)) @ (rnd(B).T if tb else rnd(B))) < 2e-5


def test_gemm_bf16_falls_back_to_fp32_for_unaligned_rows(L):
    """nn.Linear(105, h): rows of 105 floats are not 16-byte aligned -> multiplied in fp32 (documented in the header)."""
    g = torch.Generator().manual_seed(5)
    A, B = torch.randn(301, 105, generator=g).to(DEV), torch.randn(64, 105, generator=g).to(DEV)
    C = _gemm(L, L.nemo_gemm_bf16, A, B, 0, 1)
    assert rel_err(C, A.double() @ B.double().T) < 2e-5


@pytest.mark.parametrize('num_verts,N', [(700, 40), (6890, 50)])
def test_v2v_fused_bf16_blend_vs_fp32_kernel(L, num_verts, N):
    import hipops as H
    from test_gpu_ops import _ctx, _rand_rot
    assets, ctx, idx = _ctx(num_verts, 2)
    gen = torch.Generator().manual_seed(47 + N)
    R2 = _rand_rot(gen, 2 * N * 24, small=True).reshape(2 * N, 24, 3, 3)
    R2[N:, 0] = R2[:N, 0]
    Z = lambda *s: torch.zeros(*s, device=DEV)
    dR2 = H.dev(R2.reshape(2 * N, 24, 9))
    A, Jp, PF = Z(2 * N, 24, 12), Z(2 * N, 24, 3), Z(2 * N, 208)
    assert L.nemo_fk_fwd(ctx.handle, 2 * N, dR2.data_ptr(), A.data_ptr(), Jp.data_ptr(), PF.data_ptr(), 208, H.st()) == 0
    ldn = (N + 15) // 16 * 16
    out = []
    for fn in (L.nemo_v2v_fused, L.nemo_v2v_fused_bf16):
        loss, dVPt, dA = Z(1), Z(3 * ctx.NVp, ldn), Z(N, 24, 12)
        ws = torch.zeros(int(L.nemo_v2v_fused_ws_bytes(ctx.handle, N)) // 4 + 1, device=DEV)
        for _ in range(2):
            loss.zero_()
            assert fn(ctx.handle, N, PF.data_ptr(), 208, A.data_ptr(), loss.data_ptr(), dVPt.data_ptr(), ldn,
                      dA.data_ptr(), ws.data_ptr(), ws.numel() * 4, H.st()) == 0
        out.append((loss.clone(), dVPt.clone(), dA.clone()))
    (l32, v32, a32), (l16, v16, a16) = out
    assert float(l32) > 0 and rel_err(l16, l32) < 2e-3
    assert rel_err(l16, l32) > 0 or bool((a16 != a32).any())       # (it really is another arithmetic)
    # dVP = T^T sign(.) does not depend on the blend at all except through sign flips near ties
    assert float(((v16 - v32).abs() > 1e-4 * float(v32.abs().max())).float().mean()) < 2e-2
    assert rel_err(a16, a32) < 2e-2


@pytest.mark.parametrize('num_verts,N', [(700, 40), (6890, 50)])
def test_v2v_fused_bf16_split_precision_skinning(L, num_verts, N):
    """With ZERO pose features the blend contributes nothing (vp == v_shaped in both kernels), so whatever separates the
    bf16 kernel from the fp32 one is its vertex->joint adjoint and (MODE 2, the default since round 5) its skinning -- which run on
    the bf16 pipe in split precision (two bf16 pieces per fp32 operand, csrc/smpl.hip MODE 2 / 3): 16 significant bits,
    i.e. two orders below bf16."""
    import hipops as H
    from test_gpu_ops import _ctx, _rand_rot
    assets, ctx, idx = _ctx(num_verts, 2)
    gen = torch.Generator().manual_seed(11 + N)
    R2 = _rand_rot(gen, 2 * N * 24, small=True).reshape(2 * N, 24, 3, 3)
    Z = lambda *s: torch.zeros(*s, device=DEV)
    dR2 = H.dev(R2.reshape(2 * N, 24, 9))
    A, Jp, PF = Z(2 * N, 24, 12), Z(2 * N, 24, 3), Z(2 * N, 208)
    assert L.nemo_fk_fwd(ctx.handle, 2 * N, dR2.data_ptr(), A.data_ptr(), Jp.data_ptr(), PF.data_ptr(), 208, H.st()) == 0
    PF.zero_()
    ldn = (N + 15) // 16 * 16
    out = []
    for fn in (L.nemo_v2v_fused, L.nemo_v2v_fused_bf16):
        loss, dVPt, dA = Z(1), Z(3 * ctx.NVp, ldn), Z(N, 24, 12)
        ws = torch.zeros(int(L.nemo_v2v_fused_ws_bytes(ctx.handle, N)) // 4 + 1, device=DEV)
        loss.zero_()
        assert fn(ctx.handle, N, PF.data_ptr(), 208, A.data_ptr(), loss.data_ptr(), dVPt.data_ptr(), ldn,
                  dA.data_ptr(), ws.data_ptr(), ws.numel() * 4, H.st()) == 0
        out.append((loss.clone(), dVPt.clone(), dA.clone()))
    (l32, v32, a32), (l16, v16, a16) = out
    assert float(l32) > 0 and rel_err(l16, l32) < 2e-5
    assert float(((v16 - v32).abs() > 1e-4 * float(v32.abs().max())).float().mean()) < 1e-3      # (sign flips at ties only)
    # dA sums g [vp; 1] over the vertices: ONE sign that flips at an L1 tie moves an entry by 2 w |vp| (3.5e-2 of the tensor's
    # scale seen at 700 vertices).  Samples with a coordinate within 8e-6 of a tie -- the split-precision skinning's error is
    # 4e-6 of the transforms -- are compared by their loss and d vp above, not by dA.
    from test_gpu_ops import _mesh_term_f64
    d = _mesh_term_f64(assets, PF, A, N)[3]
    clean = d.abs().reshape(N, -1).min(1).values > 8e-6
    assert int(clean.sum()) >= 5                                        # (20 670 coordinates per sample: 15 of 50 samples at 6890 vertices)
    assert rel_err(a16[clean], a32[clean]) < 1e-3
    assert bool((a16 != a32).any())                                     # (it really is another arithmetic)


def test_unknown_gemm_dtype_is_rejected():
    from nemo_cvpr2023_amd.neural_motion_model import NemoV2
    args = syn.published_args(h_dim=48, monotonic_network_n_nodes=20, batch_size=8, out_dir='', phase_rbf_dim=16)
    args.gemm_dtype = 'fp8'
    with pytest.raises(ValueError):
        NemoV2(args, syn.SyntheticSequences(2, 4, seed=1), DEV, smpl_assets=syn.make_smpl_assets(128, seed=1),
               vposer_state=syn.make_vposer_state(), gmm=syn.make_gmm())


@pytest.mark.parametrize('skin_nnz', [4, 24])
def test_c3_bf16_step_vs_fp32_oracle(skin_nnz):
    """BASELINE configs[2] at its real size: 40 instances x 300 frames, h = 1000, 6890 vertices, every loss term,
    gemm_dtype = 'bf16'; the body model with SMPL's 4 non-zero skinning weights per vertex (sparse-skinning mesh kernel) and
    with a dense weight matrix.  Rows of ~300 samples against the fp32 oracle (bf16 tolerance), the full-batch losses against
    the fp32 HIP path from the same state, MLP gradients, and three descending update steps."""
    from nemo_cvpr2023_amd.neural_motion_model import NemoV2
    from oracle.model import OracleNemo
    V, T = 40, 300
    seqs = syn.SyntheticSequences(V, T, seed=1234)
    assets, vps, gmm = syn.make_smpl_assets(6890, seed=1, skin_nnz=skin_nnz), syn.make_vposer_state(), syn.make_gmm()
    models = {}
    for dt in ('f32', 'bf16'):
        args = syn.published_args(batch_size=512, out_dir='')
        args.gemm_dtype = dt
        torch.manual_seed(0)
        m = NemoV2(args, seqs, DEV, smpl_assets=assets, vposer_state=vps, gmm=gmm)
        with torch.no_grad():
            m.learned_motion.rot_out.weight.mul_(2e3)
        models[dt] = m
    m32, m16 = models['f32'], models['bf16']
    assert m16.engine.bf16 and not m32.engine.bf16
    ld32, i32 = m32.step(None, None, update=False, full_batch=True)
    ld16, i16 = m16.step(None, None, update=False, full_batch=True)
    for k in ld32:
        assert rel_err(ld16[k], ld32[k]) < 5e-3, (k, ld16[k], ld32[k])
    assert rel_err(i16['j'], i32['j']) < 5e-3 and rel_err(i16['loss_all'], i32['loss_all']) < 5e-2
    assert rel_err(i16['j'], i32['j']) > 1e-6            # the MLP really ran in another arithmetic
    rows = torch.randint(0, V * T, (300,), generator=torch.Generator().manual_seed(3))
    o = OracleNemo(2, m16.args, seqs, assets, vps, gmm, state={k: v.detach().cpu() for k, v in m16.state_dict().items()})
    _, io = o.step(rows // T, rows % T, update=False)
    assert rel_err(i16['j'][rows.to(DEV)], io['j']) < 5e-3
    for o_ in m16.optimizers + m32.optimizers:
        o_.param_groups[0]['lr'] = 0.0
    m32.step(None, None, update=True, full_batch=True)
    m16.step(None, None, update=True, full_batch=True)
    n32, n16 = dict(m32.named_parameters()), dict(m16.named_parameters())
    for k in ('learned_motion.net.net.2.weight', 'learned_motion.rot_out.weight', 'learned_motion.net.net.0.weight',
              'learned_cameras'):
        a, b = n16[k].grad.double().flatten(), n32[k].grad.double().flatten()
        assert float(a @ b / (a.norm() * b.norm())) > 0.995, k          # same direction ...
        assert rel_err(n16[k].grad, n32[k].grad) < 0.15, k              # ... and no entry off by more than bf16 noise
    for o_, lr in zip(m16.optimizers, (0.1, 1e-4, 1e-4, 1e-3)):
        o_.param_groups[0]['lr'] = lr
    first = float(m16.step(None, None, update=True, full_batch=True)[0]['total_loss'])
    for _ in range(4):
        last = m16.step(None, None, update=True, full_batch=True)[0]
    assert np.isfinite(float(last['total_loss'])) and float(last['total_loss']) < first


@pytest.mark.parametrize('M,N,K', [(301, 1000, 1000), (2401, 147, 1000), (130, 70, 101), (12000, 512, 63), (77, 207, 2070)])
def test_gemm_with_bf16_operands_in_memory(L, M, N, K):
    """nemo_gemm_bf16mem (round 3): operands cast once (nemo_cast_bf16, plain and transposed), then C = A B^T with half the
    bytes through the LDS-DMA.  Exactly the fp32-accumulated product of the bf16-ROUNDED operands (2e-5), i.e. what
    nemo_gemm_bf16 computes; fused bias + ReLU, C += mode, and the two bf16 copies of the result the epilogue can write (the
    result rounded to bf16, plain and transposed); ragged M / N / odd K (the cast's zero pad)."""
    import hipops as H
    from nemo_cvpr2023_amd._lib import check, dptr
    g = torch.Generator().manual_seed(M + 3 * N + 7 * K)
    A = torch.randn(M, K + 3, generator=g).to(DEV)[:, :K]
    Bt = torch.randn(K, N + 1, generator=g).to(DEV)[:, :N]                # B given as (K x N): goes through the TRANSPOSING cast
    bias = torch.randn(N, generator=g).to(DEV)
    Kp = (K + 7) // 8 * 8
    Ab = torch.full((M, Kp), 0x7fc0, dtype=torch.int16, device=DEV)       # NaN-poisoned: the cast must fill the k-pad with zeros
    Bb = torch.full((N, Kp), 0x7fc0, dtype=torch.int16, device=DEV)
    check(L.nemo_cast_bf16(M, K, dptr(A), A.stride(0), dptr(Ab), Kp, 0, H.st()), 'cast')
    check(L.nemo_cast_bf16(K, N, dptr(Bt), Bt.stride(0), dptr(Bb), Kp, 1, H.st()), 'cast T')
    rnd = lambda x: x.to(torch.bfloat16)
    assert torch.equal(Ab[:, :K].view(torch.bfloat16), rnd(A)) and torch.equal(Bb[:, :K].view(torch.bfloat16), rnd(Bt.T))
    assert int(Ab[:, K:].abs().sum()) == 0 and int(Bb[:, K:].abs().sum()) == 0
    Keven = (K + 1) // 2 * 2
    ws = H.gemm_ws()
    C = torch.zeros(M, N, device=DEV)
    Mp = (M + 7) // 8 * 8
    Cb = torch.zeros(M, N, dtype=torch.int16, device=DEV)
    CbT = torch.zeros(N, Mp, dtype=torch.int16, device=DEV)
    R = int(L.nemo_gemm_colsum_rows(M))
    assert R == 2 * ((M + 63) // 64)
    cs = torch.full((R, N + 3), float('nan'), device=DEV)           # per-band column sums of the result (bias gradients)
    check(L.nemo_gemm_bf16mem(M, N, Keven, dptr(Ab), Kp, dptr(Bb), Kp, dptr(C), N, dptr(bias), 1, None, 0, 0, 1.0, 0,
                              dptr(Cb), N, dptr(CbT), Mp, dptr(cs), N + 3, dptr(ws), ws.numel() * 4, H.st()), 'gemm_bf16mem')
    ref = torch.relu(rnd(A).double() @ rnd(Bt).double() + bias.double())
    assert rel_err(C, ref) < 2e-5
    assert bool(torch.isfinite(cs[:, :N]).all()) and bool(torch.isnan(cs[:, N:]).all())
    assert rel_err(cs[:, :N].double().sum(0), C.double().sum(0)) < 1e-5
    band = torch.zeros(R * 32, N, dtype=torch.float64, device=DEV)
    band[:M] = C.double()
    assert rel_err(cs[:, :N], band.reshape(R, 32, N).sum(1)) < 1e-5          # band b = rows [32 b, 32 b + 32)
    # without C: only the bf16 copies and the column sums are produced
    Cb2, cs2 = torch.zeros_like(Cb), torch.zeros_like(cs)
    check(L.nemo_gemm_bf16mem(M, N, Keven, dptr(Ab), Kp, dptr(Bb), Kp, None, 0, dptr(bias), 1, None, 0, 0, 1.0, 0,
                              dptr(Cb2), N, None, 0, dptr(cs2), N + 3, dptr(ws), ws.numel() * 4, H.st()), 'gemm_bf16mem')
    assert torch.equal(Cb2, Cb) and torch.equal(cs2[:, :N], cs[:, :N])
    assert torch.equal(Cb.view(torch.bfloat16), rnd(C)) and torch.equal(CbT[:, :M].view(torch.bfloat16), rnd(C).T.contiguous())
    assert int(CbT[:, M:].abs().sum()) == 0
    # the on-the-fly kernel on the fp32 operands computes the same thing (it needs 16-byte aligned rows to take its bf16 path)
    if K % 4 == 0 and N % 4 == 0:
        C1 = _gemm(L, L.nemo_gemm_bf16, A.contiguous(), Bt.contiguous(), 0, 0, bias=bias, act=1)
        assert rel_err(C, C1) < 2e-5
    # C += (parameter-gradient mode)
    C0 = torch.randn(M, N, generator=g).to(DEV)
    C2 = C0.clone()
    check(L.nemo_gemm_bf16mem(M, N, Keven, dptr(Ab), Kp, dptr(Bb), Kp, dptr(C2), N, None, 0, None, 0, 0, 1.0, 1,
                              None, 0, None, 0, None, 0, dptr(ws), ws.numel() * 4, H.st()), 'gemm_bf16mem')
    assert rel_err(C2, C0.double() + rnd(A).double() @ rnd(Bt).double()) < 2e-5
    assert int(ws[:4096].view(torch.int32).abs().sum()) == 0


@pytest.mark.parametrize('M,N,K', [(12001, 1000, 1000), (4801, 1000, 152), (3100, 778, 203), (8193, 1000, 1000), (3072, 768, 64)])
def test_large_tile_bf16_product(L, M, N, K):
    """The large-tile kernel of round 5 (csrc/gemm_b16x.h: 8 MFMA waves on a 192 x 256 / 128 x 256 tile + 4 loader waves; what
    nemo_gemm_bf16mem runs from 3072 rows on) against the fp32-accumulated product of the same bf16 operands: forward form
    (bias, ReLU, both bf16 copies), dX form (bf16 ReLU mask incl. -0 / 0 / negative entries, copies, per-band column sums),
    fp32 result in store and += mode; ragged M / N / K incl. odd K behind NaN-poisoned row pads (the kernel masks the end of K
    per element); pads of the copies zero, nothing behind them written; twice the same bits."""
    import hipops as H
    from nemo_cvpr2023_amd._lib import check, dptr
    g = torch.Generator().manual_seed(11 * M + 3 * N + 7 * K)
    r8 = lambda n: (n + 7) // 8 * 8
    A32, B32 = torch.randn(M, K, generator=g), torch.randn(N, K, generator=g)
    Ab = torch.full((M, r8(K) + 8), 0x7fc0, dtype=torch.int16)            # everything behind column K is NaN
    Bb = torch.full((N, r8(K) + 8), 0x7fc0, dtype=torch.int16)
    Ab[:, :K] = A32.to(torch.bfloat16).view(torch.int16)
    Bb[:, :K] = B32.to(torch.bfloat16).view(torch.int16)
    Ab, Bb = Ab.to(DEV), Bb.to(DEV)
    a, b = A32.to(torch.bfloat16).double().to(DEV), B32.to(torch.bfloat16).double().to(DEV)
    bias = torch.randn(N, generator=g).to(DEV)
    mk = torch.randn(M, r8(N), generator=g)
    mk[torch.rand(M, r8(N), generator=g) < 0.2] = 0.0
    mk[torch.rand(M, r8(N), generator=g) < 0.1] = -0.0
    mask = mk.to(torch.bfloat16).view(torch.int16).to(DEV)
    ws = H.gemm_ws()
    Keven = K                                                               # (odd K allowed by this kernel)
    if K % 2:
        Keven = K + 1                                                       # the ABI takes K in pairs: the pair's pad must be zero
        Ab[:, K] = 0
        Bb[:, K] = 0
    ldcb, ldcbt = r8(N) + 8, r8(M) + 16
    R = int(L.nemo_gemm_colsum_rows(M))

    def run(C, out_mode, bias_, act, mask_, want_copies, want_cs):
        Cb = torch.full((M, ldcb), 0x1111, dtype=torch.int16, device=DEV) if want_copies else None
        CbT = torch.full((N, ldcbt), 0x1111, dtype=torch.int16, device=DEV) if want_copies else None
        cs = torch.full((R, N + 3), float('nan'), device=DEV) if want_cs else None
        check(L.nemo_gemm_bf16mem(M, N, Keven, dptr(Ab), Ab.stride(0), dptr(Bb), Bb.stride(0), dptr(C), C.stride(0) if C is not None else 0,
                                  dptr(bias_), act, dptr(mask_), mask_.stride(0) if mask_ is not None else 0, 17 if mask_ is not None else 0,
                                  0.5, out_mode, dptr(Cb), ldcb, dptr(CbT), ldcbt, dptr(cs), N + 3, dptr(ws), ws.numel() * 4, H.st()),
              'gemm_bf16mem')
        return Cb, CbT, cs

    prod = 0.5 * (a @ b.T)
    # forward form
    C = torch.full((M, N + 5), float('nan'), device=DEV)
    Cb, CbT, _ = run(C[:, :N], 0, bias, 1, None, True, False)
    ref = torch.relu(prod + bias.double())
    assert rel_err(C[:, :N], ref) < 2e-5 and bool(torch.isnan(C[:, N:]).all())
    assert torch.equal(Cb[:, :N].view(torch.bfloat16), C[:, :N].to(torch.bfloat16))
    assert torch.equal(CbT[:, :M].view(torch.bfloat16), C[:, :N].to(torch.bfloat16).T.contiguous())
    assert int(Cb[:, N:r8(N)].abs().sum()) == 0 and bool((Cb[:, r8(N):] == 0x1111).all())
    assert int(CbT[:, M:r8(M)].abs().sum()) == 0 and bool((CbT[:, r8(M):] == 0x1111).all())
    Cb2, CbT2, _ = run(None, 0, bias, 1, None, True, False)                 # copies only; bit-identical
    assert torch.equal(Cb2, Cb) and torch.equal(CbT2, CbT)
    # dX form
    C = torch.zeros(M, N, device=DEV)
    Cb, CbT, cs = run(C, 0, None, 0, mask, True, True)
    ref = prod * (mk[:, :N].to(torch.bfloat16).double().to(DEV) > 0)
    assert rel_err(C, ref) < 2e-5
    assert torch.equal(Cb[:, :N].view(torch.bfloat16), C.to(torch.bfloat16))
    assert torch.equal(CbT[:, :M].view(torch.bfloat16), C.to(torch.bfloat16).T.contiguous())
    band = torch.zeros(R * 32, N, dtype=torch.float64, device=DEV)
    band[:M] = C.double()
    assert bool(torch.isnan(cs[:, N:]).all()) and rel_err(cs[:, :N], band.reshape(R, 32, N).sum(1)) < 1e-5
    # += mode
    C0 = torch.randn(M, N, generator=g).to(DEV)
    C2 = C0.clone()
    run(C2, 1, None, 0, None, False, False)
    assert rel_err(C2, C0.double() + prod) < 2e-5
    assert int(ws[:4096].view(torch.int32).abs().sum()) == 0


@pytest.mark.parametrize('M,N,K', [(1000, 1000, 30011), (520, 1000, 24577)])
def test_large_tile_bf16_parameter_gradient_in_k_slices(L, M, N, K):
    """The same kernel on the parameter-gradient shape (K = samples, from 24 576 on): K slices across workgroups, write-through
    slabs, the last arriver sums them in slice order -- twice the same bits, tickets back at zero."""
    import hipops as H
    from nemo_cvpr2023_amd._lib import check, dptr
    g = torch.Generator().manual_seed(M + 3 * N + 7 * K)
    r8 = lambda n: (n + 7) // 8 * 8
    A32, B32 = torch.randn(M, K, generator=g), torch.randn(N, K, generator=g)
    Ab = torch.full((M, r8(K) + 8), 0x7fc0, dtype=torch.int16)
    Bb = torch.full((N, r8(K) + 8), 0x7fc0, dtype=torch.int16)
    Ab[:, :K] = A32.to(torch.bfloat16).view(torch.int16)
    Bb[:, :K] = B32.to(torch.bfloat16).view(torch.int16)
    Ab[:, K] = 0
    Bb[:, K] = 0
    Ab, Bb = Ab.to(DEV), Bb.to(DEV)
    ref = A32.to(torch.bfloat16).double().to(DEV) @ B32.to(torch.bfloat16).double().to(DEV).T
    ws = H.gemm_ws()
    outs = []
    for _ in range(2):
        C = torch.ones(M, N, device=DEV)
        check(L.nemo_gemm_bf16mem(M, N, K + 1, dptr(Ab), Ab.stride(0), dptr(Bb), Bb.stride(0), dptr(C), N, None, 0, None, 0, 0, 1.0, 1,
                                  None, 0, None, 0, None, 0, dptr(ws), ws.numel() * 4, H.st()), 'gemm_bf16mem')
        outs.append(C)
    assert rel_err(outs[0], ref + 1.0) < 2e-5
    assert torch.equal(outs[0], outs[1])
    assert int(ws[:4096].view(torch.int32).abs().sum()) == 0


@pytest.mark.parametrize('M,N,K', [(2400, 207, 20670), (8192, 207, 20670), (3808, 207, 20670), (300, 207, 20670), (257, 200, 4098),
                                   (1000, 130, 2050)])
def test_blend_shape_adjoint_with_bf16_operands_on_the_mixed_shape_tile(L, M, N, K):
    """csrc/gemm_adj.h B16 (round 4): the blend-shape adjoint of the bf16-in-memory chain -- dPF (+)= dVP P^T, both operands bf16
    and k-contiguous, 128 < N <= 208 -- on ONE 64 x 208 column tile per workgroup (32x32x16 bf16 MFMAs for columns [0, 192),
    16x16x32 for the remainder) with K slices combined in the launch: the fp32-accumulated product of the bf16 operands
    (2e-5), NaN-poisoned k-pads and row pads that must never be read, accumulate mode, run-to-run bit equality, tickets back
    at zero."""
    import hipops as H
    from nemo_cvpr2023_amd._lib import check, dptr
    g = torch.Generator().manual_seed(M + 3 * N + 7 * K)
    Kp = (K + 7) // 8 * 8 + 8
    A = torch.randn(M, K, generator=g).to(DEV).to(torch.bfloat16)
    B = torch.randn(N, K, generator=g).to(DEV).to(torch.bfloat16)
    Ab = torch.full((M + 5, Kp), 0x7fc0, dtype=torch.int16, device=DEV)      # NaN beyond K and beyond M
    Bb = torch.full((N + 3, Kp), 0x7fc0, dtype=torch.int16, device=DEV)
    Ab[:M, :K] = A.view(torch.int16)
    Bb[:N, :K] = B.view(torch.int16)
    ref = A.double() @ B.double().T
    ws = H.gemm_ws()

    def run(C, alpha, mode):
        check(L.nemo_gemm_bf16mem(M, N, K, dptr(Ab), Kp, dptr(Bb), Kp, dptr(C), 208, None, 0, None, 0, 0, alpha, mode,
                                  None, 0, None, 0, None, 0, dptr(ws), ws.numel() * 4, H.st()), 'gemm_bf16mem')
        torch.cuda.synchronize()
        return C
    C0 = torch.randn(M, 208, generator=g).to(DEV)
    C1 = run(C0.clone(), 1.0, 0)
    assert rel_err(C1[:, :N], ref) < 2e-5
    assert torch.equal(C1[:, N:], C0[:, N:])                           # columns beyond N untouched
    assert torch.equal(C1, run(C0.clone(), 1.0, 0))
    C2 = run(C0.clone(), 0.5, 1)
    assert rel_err(C2[:, :N], C0[:, :N].double() + 0.5 * ref) < 2e-5
    assert int(ws[:4096].view(torch.int32).abs().sum()) == 0


@pytest.mark.parametrize('M,N,K', [(301, 1000, 105), (2401, 1000, 105), (130, 70, 33), (77, 200, 64)])
def test_fp32_product_with_bf16_output_copies(L, M, N, K):
    """nemo_gemm_f32_b16out: the first MotionNet layer of the bf16-in-memory chain -- fp32 operands (rows of 105 floats),
    fp32 arithmetic, the result stored as bf16 and as its bf16 transpose straight from the epilogue (and in fp32 when asked)."""
    import hipops as H
    from nemo_cvpr2023_amd._lib import check, dptr
    g = torch.Generator().manual_seed(M + N + K)
    A, B, bias = torch.randn(M, K, generator=g).to(DEV), torch.randn(N, K, generator=g).to(DEV), torch.randn(N, generator=g).to(DEV)
    ws = H.gemm_ws()
    Mp, Np = (M + 7) // 8 * 8, (N + 7) // 8 * 8
    for with_c in (True, False):
        C = torch.zeros(M, N, device=DEV)
        Cb = torch.zeros(M, Np, dtype=torch.int16, device=DEV)
        CbT = torch.zeros(N, Mp, dtype=torch.int16, device=DEV)
        check(L.nemo_gemm_f32_b16out(0, 1, M, N, K, dptr(A), K, dptr(B), K, dptr(C) if with_c else None, N, dptr(bias), 1,
                                     dptr(Cb), Np, dptr(CbT), Mp, dptr(ws), ws.numel() * 4, H.st()), 'gemm_f32_b16out')
        ref = torch.relu(A.double() @ B.double().T + bias.double())
        if with_c:
            assert rel_err(C, ref) < 2e-5
            assert torch.equal(Cb[:, :N].view(torch.bfloat16), C.to(torch.bfloat16))
        assert rel_err(Cb[:, :N].view(torch.bfloat16).double(), ref) < 5e-3
        assert torch.equal(CbT[:, :M].view(torch.bfloat16), Cb[:, :N].view(torch.bfloat16).T.contiguous())
        assert int(CbT[:, M:].abs().sum()) == 0 and int(Cb[:, N:].abs().sum()) == 0
    assert L.nemo_gemm_f32_b16out(0, 1, M, N, K, dptr(A), K, dptr(B), K, None, 0, None, 0, None, 0, None, 0, dptr(ws),
                                  ws.numel() * 4, H.st()) != 0                   # neither copy asked for
