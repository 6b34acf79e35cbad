"""A/B of the fused mesh kernel: dense 24-joint skinning product against the sparse (<= 4 non-zero weights per vertex)
form, on the SAME 4-sparse weights.  Prints the differences of loss / d vp / dA and the launch times.
usage: python tools/debug/mesh_sparse_ab.py [N] [bf16]"""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from nemo_cvpr2023_amd import _lib, synthetic as syn
from nemo_cvpr2023_amd.engine import SmplContext

N = int(sys.argv[1]) if len(sys.argv) > 1 else 2400
bf16 = len(sys.argv) > 2 and sys.argv[2] == 'bf16'
L = _lib.load()
assets = syn.make_smpl_assets(6890, seed=1, skin_nnz=4)
jm = [int(x) for x in assets['joint_map']]
ctx = SmplContext(assets, [jm[i] for i in [38] + list(range(1, 25))], 'cuda:0')
print('skin_nnz', ctx.skin_nnz, 'sparse', ctx.skin_sparse)
g = torch.Generator().manual_seed(3)
th = 0.3 * torch.randn(2 * N * 24, 3, generator=g)
ang = th.norm(dim=1, keepdim=True).clamp_min(1e-8)
ax = th / ang
K = torch.zeros(2 * N * 24, 3, 3)
K[:, 0, 1], K[:, 0, 2], K[:, 1, 0], K[:, 1, 2], K[:, 2, 0], K[:, 2, 1] = -ax[:, 2], ax[:, 1], ax[:, 2], -ax[:, 0], -ax[:, 1], ax[:, 0]
R = torch.eye(3) + ang.sin()[..., None] * K + (1 - ang.cos())[..., None] * (K @ K)
R2 = R.reshape(2 * N, 24, 9).cuda()
st = torch.cuda.current_stream().cuda_stream
Z = lambda *s: torch.zeros(*s, device='cuda')
A, Jp, PF = Z(2 * N, 24, 12), Z(2 * N, 24, 3), Z(2 * N, 208)
assert L.nemo_fk_fwd(ctx.handle, 2 * N, R2.data_ptr(), A.data_ptr(), Jp.data_ptr(), PF.data_ptr(), 208, st) == 0
ldn = (N + 15) // 16 * 16
ws = torch.zeros(int(L.nemo_v2v_fused_ws_bytes(ctx.handle, N)) // 4 + 1, device='cuda')
fn = L.nemo_v2v_fused_bf16 if bf16 else L.nemo_v2v_fused
out = {}
for sparse in (False, True):
    ctx.set_skin_sparse(sparse)
    loss, dVPt, dA = Z(1), Z(3 * ctx.NVp, ldn), Z(N, 24, 12)
    def run():
        assert fn(ctx.handle, N, PF.data_ptr(), 208, A.data_ptr(), loss.data_ptr(), dVPt.data_ptr(), ldn,
                  dA.data_ptr(), ws.data_ptr(), ws.numel() * 4, st) == 0
    for _ in range(3):
        loss.zero_(); run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 20
    e0.record()
    for _ in range(reps):
        run()
    e1.record(); torch.cuda.synchronize()
    loss.zero_(); run(); torch.cuda.synchronize()
    out[sparse] = (loss.clone(), dVPt.clone(), dA.clone())
    print('sparse' if sparse else 'dense ', f'{e0.elapsed_time(e1) / reps * 1e3:8.1f} us/launch', 'loss', float(loss))
rel = lambda a, b: float((a - b).norm() / b.norm())
print('loss rel', rel(out[True][0], out[False][0]), 'dVPt rel', rel(out[True][1], out[False][1]), 'dA rel',
      rel(out[True][2], out[False][2]))
