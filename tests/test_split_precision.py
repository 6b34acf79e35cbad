"""The arithmetic behind `mesh_blend = 'f32_split'` (csrc/smpl.hip MODE 5 / 4), restated in torch on the CPU: fp32 operands carried
as two fp16 pieces of s x (or three bf16 pieces), the piece products that matter summed in fp32 -- against float64 and against a
plain fp32 product of the same operands.  (The kernel itself is held to the fp32-MFMA kernel's error on the GPU:
tests/test_gpu_ops.py::test_v2v_fused_split_is_fp32_equivalent.)"""
import torch


def _pieces(x, n, dtype, scale=1.0):
    """x (fp32) -> n pieces of `dtype` whose sum is scale * x up to the last piece's rounding."""
    r = (x * scale).float()
    out = []
    for _ in range(n):
        h = r.to(dtype)
        out.append(h)
        r = r - h.float()          # exact in fp32: the remainder of a rounding to fewer bits
    return out


def _split_matmul(P, pf, n, dtype, sP=1.0, spf=1.0):
    """sum over the piece pairs (i, j) with i + j < n of P_i pf_j^T, each product and the sum in fp32 (the MFMA's accumulation)."""
    Pp, qp = _pieces(P, n, dtype, sP), _pieces(pf, n, dtype, spf)
    acc = torch.zeros(P.shape[0], pf.shape[0], dtype=torch.float32)
    for i in range(n):
        for j in range(n - i):
            acc = acc + Pp[i].float() @ qp[j].float().t()      # (fp32 matmul of 11-bit / 8-bit operands: products exact)
    return acc / (sP * spf)


def test_two_fp16_pieces_carry_an_fp32_value_to_one_ulp():
    """11 + 11 significant bits and the remainder's sign: 23 of the 24 bits in the worst case -- an fp32 value to 2^-23 (one ulp);
    three bf16 pieces (8 + 8 + 8) carry all 24."""
    g = torch.Generator().manual_seed(0)
    x = torch.randn(200000, generator=g) * torch.logspace(-5, 1, 200000)      # six decades of magnitudes
    s = 2.0 ** (13 - int(torch.log2(x.abs().max()).floor()))                  # max |x| s in [2^13, 2^14)
    p0, p1 = _pieces(x, 2, torch.float16, s)
    err = ((p0.double() + p1.double()) / s - x.double()).abs()
    normal = (x.abs() * s) >= 0.25                                            # x1 is a normal fp16 from here on
    assert float((err[normal] / x.double().abs()[normal]).max()) <= 2.0 ** -23 + 1e-12
    assert float(err[~normal].max()) <= 2.0 ** -24 / s + 1e-30                 # below: a subnormal x1 -- absolute, 2^-38 of max |x|
    b = _pieces(x, 3, torch.bfloat16)
    errb = (sum(t.double() for t in b) - x.double()).abs() / x.double().abs()
    assert float(errb.max()) <= 2.0 ** -24 + 1e-12


def test_split_products_are_as_accurate_as_the_fp32_product():
    """A 207-term blend (lbs.py:229-233) per output: error against float64 of (a) the plain fp32 product, (b) two fp16 pieces x
    three products, (c) three bf16 pieces x six products -- (b) and (c) must not exceed 1.5 x (a)."""
    g = torch.Generator().manual_seed(1)
    P = 1e-2 * torch.randn(3 * 512, 207, generator=g)           # blend shapes, [vertex coordinate][blend shape]
    pf = torch.randn(64, 207, generator=g).clamp(-2, 2)         # pose features R - I
    ref = P.double() @ pf.double().t()
    scale = float(ref.abs().max())
    e32 = float(((P @ pf.t()).double() - ref).abs().max()) / scale
    sP = 2.0 ** (13 - int(torch.log2(P.abs().max()).floor()))
    e16 = float((_split_matmul(P, pf, 2, torch.float16, sP, 4096.0).double() - ref).abs().max()) / scale
    eb16 = float((_split_matmul(P, pf, 3, torch.bfloat16).double() - ref).abs().max()) / scale
    one_piece = float((_split_matmul(P, pf, 1, torch.bfloat16).double() - ref).abs().max()) / scale
    print('max error / max |result|: fp32', e32, 'two fp16 pieces', e16, 'three bf16 pieces', eb16, 'plain bf16', one_piece)
    assert e16 <= 1.5 * e32 and eb16 <= 1.5 * e32
    assert one_piece > 100 * e32                                  # (the check can tell narrow arithmetic apart)
