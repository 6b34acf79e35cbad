"""Fit driver: the optimisation loop of ``scripts/learned_multi_view_recon_nn.py:211-335`` around the
MI355X engine.  Not the reference script: no rendering, no matplotlib, no wandb --
only what determines the *numbers* of a run:

  1. eval at init     one (view, frame) draw, ``step(update=False, full_batch=args.eval_full_batch)``  (:213-226)
  2. ``warmup(args.warmup_step)``                                                                    (:228)
  3. ``opt_cam(args.opt_cam_step)``                                                                  (:235)
  4. ``n_steps`` x: at step 0 and whenever (step+1) % 500 == 0 -> save checkpoint, draw, eval step   (:247-268);
     then draw the training batch (views first, then frames, both from the CPU global RNG, :291-296)
     and ``step``; collect every ``loss_dict`` entry and ``optimizers[i].param_groups[0]['lr']``     (:300-306)
  5. metrics: ``eval_2d``, ``eval_3d``, ``eval_3d(dynamic_only=True)``                               (:333-335)

The model is duck-typed (``NemoV1..V4`` of this package, a ``ShardedNemo``, or the CPU oracle in the
tests): the driver touches ``num_views, num_frames, step, warmup, opt_cam, save, optimizers``.

    python -m nemo_cvpr2023_amd.fit --synthetic 8x300 --n-steps 2000 --out-dir out/run0
"""
import argparse
import json
import os
import time
from collections import defaultdict

import numpy as np
import torch

LR_NAMES = ('lr_cam', 'lr_pose', 'lr_orient', 'lr_trans', 'lr_phase')      # scripts/...:303 (zip truncates)


def draw_batch(num_views, num_frames, batch_size):
    """scripts/learned_multi_view_recon_nn.py:213-218, :291-296: views first, then frames, CPU global RNG."""
    view_idx = torch.randint(0, num_views, size=(batch_size,))
    frame_idx = torch.randint(0, num_frames, size=(batch_size,))
    return view_idx, frame_idx


def _to_host(d):
    out = {}
    for k, v in d.items():
        if torch.is_tensor(v):
            v = v.detach().cpu().numpy()
        out[k] = v
    return out


def run_fit(model, args, out_dir=None, eval_every=500, log=None, evaluate=None):
    """Runs phases 1-4 (and 5 when ``evaluate`` is given: a callable ``evaluate(model, out_dir) -> dict``).
    Returns {'init': loss_dict, 'warmup_losses', 'cam_losses', 'losses': {key: [..]}, 'learning_rates',
    'evals': {step: loss_dict}, 'metrics'}."""
    V, T, B = model.num_views, model.num_frames, args.batch_size
    full = bool(getattr(args, 'eval_full_batch', False))
    if out_dir:
        for sub in ('ckpt', 'info'):
            os.makedirs(os.path.join(out_dir, sub), exist_ok=True)

    def dump(name, loss_dict, info_dict):
        if out_dir:
            torch.save({'loss_dict': _to_host(loss_dict), 'info_dict': _to_host(info_dict)},
                       os.path.join(out_dir, 'info', name))

    res = {'evals': {}}
    # 1. eval at init
    vi, fi = draw_batch(V, T, max(B, 1))
    ld, info = model.step(vi, fi, update=False, full_batch=full)
    res['init'] = dict(ld)
    dump('_init.pt', ld, info)
    # 2. / 3.
    def clock():
        if torch.cuda.is_available():
            torch.cuda.synchronize()
        return time.perf_counter()
    t_a = clock()
    res['warmup_losses'] = [float(x) for x in model.warmup(args.warmup_step)]
    t_b = clock()
    res['cam_losses'] = [float(x) for x in model.opt_cam(args.opt_cam_step)]
    t_c = clock()
    # 4.
    losses, lrs = defaultdict(list), defaultdict(list)
    t0 = time.perf_counter()
    for step_idx in range(args.n_steps):
        if step_idx == 0 or (step_idx + 1) % eval_every == 0:
            if out_dir:
                model.save(os.path.join(out_dir, 'ckpt', f'sd_{step_idx:06d}.pt'))
            vi, fi = draw_batch(V, T, max(B, 1))
            ld, info = model.step(vi, fi, update=False, full_batch=full)
            res['evals'][step_idx] = dict(ld)
            dump(f'{step_idx:06d}.pt', ld, info)
        if B > 0:
            vi, fi = draw_batch(V, T, B)
        else:
            vi = fi = None
        ld, _ = model.step(vi, fi)
        for k, v in ld.items():
            losses[k].append(float(v))
        for name, opt in zip(LR_NAMES, model.optimizers):
            lrs[name].append(float(opt.param_groups[0]['lr']))
        if log is not None:
            log(step_idx, ld)
    res['seconds'] = clock() - t0
    res['phase_seconds'] = {'warmup': t_b - t_a, 'opt_cam': t_c - t_b, 'steps': res['seconds']}
    res['losses'], res['learning_rates'] = dict(losses), dict(lrs)
    # 5.
    res['metrics'] = evaluate(model, out_dir) if evaluate is not None else None
    if out_dir:
        with open(os.path.join(out_dir, 'curves.json'), 'w') as f:
            json.dump({'warmup_losses': res['warmup_losses'], 'cam_losses': res['cam_losses'],
                       'losses': res['losses'], 'learning_rates': res['learning_rates'],
                       'seconds': res['seconds']}, f)
    return res


def main(argv=None):
    from . import synthetic as syn
    from .neural_motion_model import NEMO_VERSIONS
    ap = argparse.ArgumentParser(description=__doc__.split('\n\n')[0])
    ap.add_argument('--synthetic', default='8x300', help='VxT synthetic Baseball-Pitch-shaped sequences')
    ap.add_argument('--nemo-cfg', default='', help='nemo/config/*.yml of a NeMo-MoCap action: fit the real data '
                    '(needs the SMPL / VPoser / GMM files under software/, see assets.py)')
    ap.add_argument('--mocap-root', default='data/mocap')
    ap.add_argument('--n-frames', type=int, default=1000000)
    ap.add_argument('--model-version', type=int, default=2)
    ap.add_argument('--n-steps', type=int, default=2000)
    ap.add_argument('--warmup-step', type=int, default=300)
    ap.add_argument('--opt-cam-step', type=int, default=1000)
    ap.add_argument('--batch-size', type=int, default=512)
    ap.add_argument('--num-verts', type=int, default=6890)
    ap.add_argument('--out-dir', default='')
    ap.add_argument('--seed', type=int, default=0)
    ap.add_argument('--no-eval', action='store_true')
    o = ap.parse_args(argv)
    V, T = (int(x) for x in o.synthetic.lower().split('x'))
    base = syn.published_args if o.model_version >= 2 else syn.default_v1_args
    args = base(batch_size=o.batch_size, out_dir=o.out_dir, n_steps=o.n_steps, warmup_step=o.warmup_step,
                opt_cam_step=o.opt_cam_step)
    args.model_version = o.model_version
    torch.manual_seed(o.seed)
    if o.nemo_cfg:
        import yaml
        from .multi_view_sequence import load_nemo_mocap
        with open(o.nemo_cfg) as f:
            seqs = load_nemo_mocap(yaml.safe_load(f), 0, o.n_frames, mocap_root=o.mocap_root)
        model = NEMO_VERSIONS[o.model_version](args, seqs, 'cuda:0')       # model files from software/
    else:
        seqs = syn.SyntheticSequences(V, T, seed=1234, with_eval=True)
        model = NEMO_VERSIONS[o.model_version](
            args, seqs, 'cuda:0', smpl_assets=syn.make_smpl_assets(o.num_verts, seed=1, skin_nnz=4),   # (SMPL's skinning sparsity)
            vposer_state=syn.make_vposer_state(), gmm=syn.make_gmm())
    evaluate = None
    if not o.no_eval:
        from .evaluation import evaluate_all
        evaluate = evaluate_all
    res = run_fit(model, args, out_dir=o.out_dir or None, evaluate=evaluate,
                  log=lambda s, ld: print(s, float(ld['total_loss']), float(ld['kp_loss'])) if s % 100 == 0 else None)
    print(json.dumps({'steps': o.n_steps, 'seconds': round(res['seconds'], 3),
                      'iters_per_s': round(o.n_steps / max(res['seconds'], 1e-9), 2),
                      'final_total_loss': res['losses']['total_loss'][-1] if res['losses'] else None,
                      'metrics': res['metrics']}))


if __name__ == '__main__':
    main()
