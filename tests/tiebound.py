"""What sign(0) ties of the L1 mesh term can change in a gradient -- so that gradients with the v2v term ON are held to the
same 1e-4 as everything else instead of a flat 2e-3.

The term is  c * sum_i |v_rec_i - v_orig_i|  (v_rec detached); its gradient is  -c * sum_i sign(d_i) dv_orig_i/dtheta.  Two
correct fp32 evaluations can only disagree on sign(d_i) where d_i is within their own rounding of zero.  In a FLOAT64 evaluation
of the same state we list the coordinates with |d_i| <= tau * max|v| (tau = 2e-6: ~16 ulp of the fp32 vertices, generous for
two independent fp32 evaluations of a 24-joint chain + 207-term blend) and bound what flipping each of them changes:
2 c |dv_orig_i / dtheta|, from one float64 backward pass per listed coordinate.  No product change, no test-only kernel input."""
import torch

TAU = 2e-6


def l1_tie_bound(forward_vo, params, v_rec, coef, tau=TAU):
    """forward_vo() -> v_orig (N, NV, 3) float64 depending on `params` (list of float64 leaf tensors requiring grad).
    Returns ([bound tensor per param], number of listed coordinates)."""
    vo = forward_vo()
    d = (v_rec - vo).detach()
    thr = tau * float(torch.maximum(v_rec.abs().max(), vo.detach().abs().max()))
    idx = (d.abs() <= thr).nonzero()
    bounds = [torch.zeros_like(p) for p in params]
    for n, (s, v, c) in enumerate(idx.tolist()):
        g = torch.autograd.grad(vo[s, v, c], params, retain_graph=True, allow_unused=True)
        for b, gi in zip(bounds, g):
            if gi is not None:
                b += 2.0 * coef * gi.abs()
    return bounds, len(idx)


def model_v2v_tie_bound(o64, view_idx, frame_idx, tau=TAU, chunk=300):
    """Per-parameter bound for a whole step of an oracle model `o64` (float64 twin, tests/test_gpu_model._float64_twin) on the
    batch (view_idx, frame_idx): {name: tensor}, number of listed coordinates.  The listed coordinates are found chunk by chunk
    without autograd; each one then costs a single-sample float64 forward + backward."""
    from oracle import ops
    a = o64.args
    N = len(view_idx)
    NV = None
    ties = []
    prev = torch.get_default_dtype()
    torch.set_default_dtype(torch.float64)
    try:
        def bodies(vi, fi):
            pd = o64.get_preds_batch(vi, fi)
            poses, orient = pd['poses'], pd['orient']
            n = poses.shape[0]
            mean, _ = o64.vp.encode(poses[:, :63])
            dec_aa, _ = o64.vp.decode(mean)
            recon = torch.cat([dec_aa.reshape(n, -1), poses[:, 63:]], 1)
            R_orig = ops.batch_rodrigues(poses.reshape(-1, 3)).reshape(n, 23, 3, 3)
            R_rec = ops.batch_rodrigues(recon.reshape(-1, 3)).reshape(n, 23, 3, 3)
            return o64._smpl(R_orig, orient)[0], o64._smpl(R_rec, orient)[0]
        with torch.no_grad():
            vmax = 0.0
            ds = []
            for s0 in range(0, N, chunk):
                vo, vr = bodies(view_idx[s0:s0 + chunk], frame_idx[s0:s0 + chunk])
                NV = vo.shape[1]
                vmax = max(vmax, float(vo.abs().max()), float(vr.abs().max()))
                ds.append((vr - vo).abs())
            thr = tau * vmax
            for ci, d in enumerate(ds):
                for s, v, c in (d <= thr).nonzero().tolist():
                    ties.append((ci * chunk + s, v, c))
        coef = float(a.weight_vp_loss) / (N * NV * 3)
        bound = {k: torch.zeros_like(p) for k, p in o64.P.items()}
        names = list(o64.P)
        for s, v, c in ties:
            vo, _ = bodies(view_idx[s:s + 1], frame_idx[s:s + 1])
            g = torch.autograd.grad(vo[0, v, c], [o64.P[k] for k in names], allow_unused=True)
            for k, gi in zip(names, g):
                if gi is not None:
                    bound[k] += 2.0 * coef * gi.abs()
    finally:
        torch.set_default_dtype(prev)
    return bound, len(ties)
