"""Fit engine: the per-iteration NeMo optimisation step as an explicit sequence of HIP kernels.

Host-side orchestration of ``libnemo_hip.so`` for the hot path of the reference
(``nemo/neural_motion_model.py`` ``NemoV*.step`` :3511-3598 / :3796-3909, ``warmup`` :3455-3509,
``opt_cam`` :2869-2906 / :4060-4151).  There is no autograd here: forward, hand-derived backward
and the fused Adam update are ~60 kernel launches on one HIP stream writing into pre-allocated
workspaces, so the same sequence can be replayed / captured.  PyTorch only provides device
memory, the stream and (for multi-GPU) ``torch.distributed``.

Parameters live in ONE flat fp32 buffer laid out in optimiser order
``[cameras | motion MLP (+RBF log_sigmas) | phase networks | instance codes]``; gradients and the
Adam moments are flat buffers with the same layout.  The ``nn.Parameter`` objects the drop-in
classes expose are views into it (so ``state_dict`` keeps the reference's key names).
"""
from __future__ import annotations

import ctypes
import os
import time
import math
from collections import OrderedDict

import numpy as np
import torch

from . import _lib
from ._lib import AdamSeg, ColsumDesc, check, dptr

FOCAL_LENGTH = 5000.0          # hmr/hmr_constants.py:1
LOSS_TYPES = {'mse_robust': 0, 'mse': 1, 'rmse': 2, 'rmse_robust': 3, 'mse_robust_resized': 4,
              'rmse_resized': 5}
RBF_KERNELS = {'quadratic': 0, 'linear': 1, 'gaussian': 2, 'inverse_quadratic': 3, 'multiquadric': 4,
               'inverse_multiquadric': 5, 'spline': 6, 'poisson_one': 7, 'poisson_two': 8, 'matern32': 9,
               'matern52': 10}
# slots of the per-step device scalar buffer
S_KP, S_V2V, S_KL, S_GMM, S_3D = 0, 1, 2, 3, 4          # (5: instance-code term, 6: temporal smoothness)
HEAD_LD = 148      # row stride of the merged MLP head output [rot6d 144 | trans 3 | pad]


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


class SmplContext:
    """Owns a ``nemo_ctx`` (device constants of one SMPL model + output-joint selection)."""

    def __init__(self, assets: dict, out_joints, device):
        self.lib = _lib.load()
        torch.cuda.set_device(device)
        f = lambda k: np.ascontiguousarray(assets[k].detach().cpu().numpy().astype(np.float32))
        vt, sd, pd, jr, w = f('v_template'), f('shapedirs'), f('posedirs'), f('J_regressor'), f('lbs_weights')
        jx = f('J_regressor_extra')
        par = np.ascontiguousarray(np.asarray(assets['parents'], dtype=np.int64))
        sel = np.ascontiguousarray(np.asarray(assets['extra_vids'], dtype=np.int64))
        oj = np.ascontiguousarray(np.asarray(out_joints, dtype=np.int64))
        self.NV = vt.shape[0]
        assert pd.shape == (207, self.NV * 3) and w.shape == (self.NV, 24) and jr.shape == (24, self.NV)
        h = ctypes.c_void_p()
        P = lambda a: a.ctypes.data_as(ctypes.c_void_p)
        check(self.lib.nemo_ctx_create(ctypes.byref(h), self.NV, P(vt), P(sd), P(pd), P(jr), P(w), P(par),
                                       jx.shape[0], P(jx), len(sel), P(sel), len(oj), P(oj)),
              'nemo_ctx_create')
        self.handle = h
        self.n_out = len(oj)
        self.nq = int(self.lib.nemo_ctx_nq(h))
        self.C1 = self.lib.nemo_ctx_C1(h)
        self.c0 = self.lib.nemo_ctx_c0(h)
        self.posedirs = self.lib.nemo_ctx_posedirs(h)
        self.ldP = int(self.lib.nemo_ctx_posedirs_ld(h))
        self.NVp = self.ldP // 3
        self.v_shaped = self.lib.nemo_ctx_v_shaped(h)
        self._betas = np.zeros(10, dtype=np.float32)
        # the largest number of non-zero skinning weights of any vertex (4 for the published SMPL model): at <= 4 the
        # fused mesh kernel skins with the non-zero weights only
        self.skin_nnz = int(self.lib.nemo_ctx_skin_nnz(h))
        # which mesh-kernel variant launches bake in (part of every graph key that contains the mesh term)
        self.skin_sparse_flag = bool(self.lib.nemo_ctx_skin_sparse(h))
        # range guard of the fp16 split-precision mesh kernel (include/nemo_hip.h nemo_ctx_split_ok): False for a body model whose
        # blended vertices could reach 2^15.9 / 2^12 (other units than metres, extreme betas) -- then the three-bf16-piece form runs
        self.split_ok = bool(self.lib.nemo_ctx_split_ok(h))

    @property
    def skin_mfma(self):
        """mesh_blend 'f32_split' skins with split-precision MFMAs (kernel MODE 6) -- the library's default inside the transforms' range
        guard; NEMO_MESH_SKIN=sparse keeps the sparse VALU form."""
        return bool(self.lib.nemo_ctx_skin_mfma_ok(self.handle)) and os.environ.get('NEMO_MESH_SKIN') != 'sparse'

    @property
    def vp_bound(self):
        return float(self.lib.nemo_ctx_vp_bound(self.handle))

    @property
    def skin_sparse(self):
        return bool(self.lib.nemo_ctx_skin_sparse(self.handle))

    def set_skin_sparse(self, enable):
        """Dense 24-joint skinning product (False) or the non-zero weights only (True; needs skin_nnz <= 4)."""
        torch.cuda.synchronize()
        check(self.lib.nemo_ctx_set_skin_sparse(self.handle, int(bool(enable))), 'nemo_ctx_set_skin_sparse')
        self.skin_sparse_flag = bool(self.lib.nemo_ctx_skin_sparse(self.handle))

    def set_betas(self, betas):
        b = np.ascontiguousarray(np.asarray(betas, dtype=np.float32).reshape(-1)[:10])
        if not np.array_equal(b, self._betas):
            torch.cuda.synchronize()
            check(self.lib.nemo_ctx_set_betas(self.handle, b.ctypes.data_as(ctypes.c_void_p)),
                  'nemo_ctx_set_betas')
            self._betas = b.copy()
            self.split_ok = bool(self.lib.nemo_ctx_split_ok(self.handle))

    def __del__(self):
        try:
            if getattr(self, 'handle', None):
                self.lib.nemo_ctx_destroy(self.handle)
                self.handle = None
        except Exception:
            pass


def fold_vposer(sd: dict, device):
    """Frozen VPoser-v2 weights (vposer_model.py:69-88) with the two eval-mode BatchNorm layers
    folded into the Linear that follows them (float64 on the host), mu|logvar heads stacked."""
    g = lambda k: sd[k].detach().cpu().double()

    def bn(p):
        s = g(p + '.weight') / torch.sqrt(g(p + '.running_var') + 1e-5)
        return s, g(p + '.bias') - g(p + '.running_mean') * s
    s1, t1 = bn('encoder_net.1')
    s2, t2 = bn('encoder_net.4')
    W2, b2 = g('encoder_net.2.weight'), g('encoder_net.2.bias')
    W6, b6 = g('encoder_net.6.weight'), g('encoder_net.6.bias')
    # encoder tail: BN(512) -> Linear6 -> Linear7 -> [mu | logvar] has no activation in between
    # (vposer_model.py:74-78), so it is ONE affine map 512 -> 64 (3 GEMMs + their 3 adjoints -> 1 + 1)
    W6f, b6f = W6 * s2.unsqueeze(0), W6 @ t2 + b6
    W7, b7 = g('encoder_net.7.weight'), g('encoder_net.7.bias')
    Wm = torch.cat([g('encoder_net.8.mu.weight'), g('encoder_net.8.logvar.weight')], 0)
    bm = torch.cat([g('encoder_net.8.mu.bias'), g('encoder_net.8.logvar.bias')], 0)
    out = {
        'e2w': W2 * s1.unsqueeze(0), 'e2b': W2 @ t1 + b2,
        'emw': Wm @ W7 @ W6f, 'emb': Wm @ (W7 @ b6f + b7) + bm,
        'd0w': g('decoder_net.0.weight'), 'd0b': g('decoder_net.0.bias'),
        'd3w': g('decoder_net.3.weight'), 'd3b': g('decoder_net.3.bias'),
        'd5w': g('decoder_net.5.weight'), 'd5b': g('decoder_net.5.bias'),
    }
    out['e2w_p'] = torch.cat([out['e2w'], torch.zeros(out['e2w'].shape[0], 1, dtype=out['e2w'].dtype)], 1)   # ld 64: 16-byte rows
    # decode(q_z.mean) (:3569, vposer_model.py:100-113): the first decoder layer reads the mean head's output with nothing in
    # between, so it is ONE affine map of the encoder's hidden activation (512 -> 512, rank 32) -- the decoder chain then
    # starts beside the (mu | logvar) product instead of behind it
    out['d0mw'] = out['d0w'] @ out['emw'][:32]
    out['d0mb'] = out['d0w'] @ out['emb'][:32] + out['d0b']
    return {k: v.float().contiguous().to(device) for k, v in out.items()}


def gmm_constants(gmm: dict, device):
    """hmr/smplify/prior.py:124-160: precisions = inv(cov) in fp32, merged nll weights."""
    means = gmm['means'].astype(np.float32)
    covs = gmm['covars'].astype(np.float32)
    prec = np.stack([np.linalg.inv(c) for c in covs]).astype(np.float32)
    prec = 0.5 * (prec + prec.transpose(0, 2, 1))      # exact symmetry (nemo_gmm_fwd_bwd's contract)
    sqrdets = np.array([np.sqrt(np.linalg.det(c)) for c in gmm['covars']])
    const = (2 * np.pi) ** (69 / 2.0)
    nllw = np.asarray(gmm['weights'] / (const * (sqrdets / sqrdets.min()))).astype(np.float32)
    t = lambda a: torch.tensor(np.ascontiguousarray(a), dtype=torch.float32, device=device)
    return {'means': t(means), 'prec': t(prec), 'log_nllw': torch.log(t(nllw)), 'M': means.shape[0]}


class ParamLayout:
    """Flat parameter buffer.  ``groups`` lists the tensors of each optimiser in the reference's
    parameter order (what ``state_dict`` / torch-format optimiser states follow); ``entries`` maps
    state_dict name -> (offset, shape).  Memory order differs from the optimiser order in one place:
    the rot_out / linear_out weights are adjacent and so are their biases, so the two MLP heads run as
    ONE (h -> 147) GEMM forward and backward."""

    def __init__(self, V, K, D, C, h, din, version=2):
        self.entries = OrderedDict()
        self.groups = OrderedDict()    # optimiser name -> [names] in optimiser order
        lm = 'learned_motion.'
        shapes = OrderedDict()
        order = []                     # memory order

        def reg(group, name, shape, place=True):
            shapes[name] = tuple(shape)
            self.groups.setdefault(group, []).append(name)
            if place:
                order.append(name)
        reg('cameras', 'learned_cameras', (V, 9))
        if version == 0:
            # NemoV0 (:3148-3206): RotNet poses / RotNet orient / FCNN translation on the bare warped phase, one
            # optimiser each; no instance code, no RBF
            for net, nout in (('learned_poses', 138), ('learned_orient', 6)):
                for lname, (fo, fi) in (('net.net.0', (h, 1)), ('net.net.2', (h, h)), ('net.net.4', (h, h)),
                                        ('linear', (nout, h))):
                    reg(net[len('learned_'):], f'{net}.{lname}.weight', (fo, fi))
                    reg(net[len('learned_'):], f'{net}.{lname}.bias', (fo,))
            for lname, (fo, fi) in (('net.0', (h, 1)), ('net.2', (h, h)), ('net.4', (3, h))):
                reg('trans', f'learned_trans.{lname}.weight', (fo, fi))
                reg('trans', f'learned_trans.{lname}.bias', (fo,))
            reg('comm', '_comm_scalars', (8,))
            for i in range(V):
                reg('phase', f'phase_networks.{i}.shifts', (K,))
                reg('phase', f'phase_networks.{i}.scales', (K,))
            off = 0
            for name in order:
                off = (off + 3) // 4 * 4
                self.entries[name] = (off, shapes[name])
                off += int(np.prod(shapes[name]))
            self.total = (off + 3) // 4 * 4
            return
        # Memory order of the shared (all-reduced) block = the order in which the backward FINISHES its gradients, last
        # first: [log_sigmas | 8 comm scalars | layer 0 | layer 2 | layer 4 | heads].  The three gradient buckets of the
        # overlapped all-reduce (dist.py) are then contiguous slices: heads + layer 4 (ready first), layer 2, and
        # layer 0 + RBF widths + the loss scalars (ready last) -- see `buckets()`.
        front = len(order)
        for lname, (fo, fi) in (('net.net.0', (h, din)), ('net.net.2', (h, h)), ('net.net.4', (h, h))):
            reg('motion', f'{lm}{lname}.weight', (fo, fi))
            reg('motion', f'{lm}{lname}.bias', (fo,))
        for lname, fo in (('rot_out', 144), ('linear_out', 3)):
            reg('motion', f'{lm}{lname}.weight', (fo, h), place=False)
            reg('motion', f'{lm}{lname}.bias', (fo,), place=False)
        order += [lm + 'rot_out.weight', lm + 'linear_out.weight', lm + 'rot_out.bias', lm + 'linear_out.bias']
        if D > 0:
            reg('motion', 'phase_rbf.log_sigmas', (D,), place=False)
            order.insert(front, 'phase_rbf.log_sigmas')
            front += 1
        # 8 floats that travel with the shared-gradient all-reduce (loss scalars); no optimiser owns them
        reg('comm', '_comm_scalars', (8,), place=False)
        order.insert(front, '_comm_scalars')
        for i in range(V):
            reg('phase', f'phase_networks.{i}.shifts', (K,))
            reg('phase', f'phase_networks.{i}.scales', (K,))
        if C > 0:
            reg('instance', 'learned_instance_code', (V, C))
        # every tensor starts on a 16-byte boundary (pads stay zero: zero gradient, zero Adam update), so
        # the weight matrices qualify for the GEMM's dwordx4 staging whatever V is (9 camera floats per
        # view would otherwise misalign every layer of a one-view shard)
        off = 0
        for name in order:
            off = (off + 3) // 4 * 4
            self.entries[name] = (off, shapes[name])
            off += int(np.prod(shapes[name]))
        self.total = (off + 3) // 4 * 4

    def buckets(self):
        """Contiguous [start, end) slices of the shared gradient block in the order the backward completes them."""
        lm = 'learned_motion.'
        first = [lm + 'net.net.4.weight', lm + 'net.net.4.bias', lm + 'rot_out.weight', lm + 'linear_out.weight',
                 lm + 'rot_out.bias', lm + 'linear_out.bias']
        mid = [lm + 'net.net.2.weight', lm + 'net.net.2.bias']
        last = [n for n in ('phase_rbf.log_sigmas',) if n in self.entries] + \
               ['_comm_scalars', lm + 'net.net.0.weight', lm + 'net.net.0.bias']
        out = [self.span(first), self.span(mid), self.span(last)]
        a, b = self.span(self.groups['motion'] + self.groups['comm'])
        pad = lambda x: (x + 3) // 4 * 4
        assert out[2][0] == a and pad(out[2][1]) == out[1][0] and pad(out[1][1]) == out[0][0] and out[0][1] == b, out
        return out

    def span(self, names):
        a = min(self.entries[n][0] for n in names)
        b = max(self.entries[n][0] + int(np.prod(self.entries[n][1])) for n in names)
        return a, b


class FitEngine:
    def __init__(self, version, args, V, T, img_d0, img_d1, assets, vposer_sd, gmm, device,
                 targets, gt_size, hmr_theta, hmr_mask):
        if not torch.cuda.is_available():
            raise _lib.NemoHipError('FitEngine needs an MI355X (no CPU fallback by design)')
        self.lib = _lib.load_for_engine()
        self.version, self.args = version, args
        self.device = torch.device(device)
        torch.cuda.set_device(self.device)
        self.V, self.T = V, T
        self.K = args.monotonic_network_n_nodes
        self.C = args.instance_code_size if version >= 1 else 0
        self.D = args.phase_rbf_dim if version >= 2 else 0
        self.h = args.h_dim
        # 'bf16' (BASELINE configs[2]): the dense contractions -- MotionNet / VPoser linear layers forward and backward,
        # the pose blend of the mesh term and its adjoint -- run on the bf16 matrix cores with fp32 accumulation, fp32
        # master weights and fp32 activations in memory.  Default 'f32': the reference's arithmetic (1e-4 parity gate).
        dt = getattr(args, 'gemm_dtype', 'f32') or 'f32'
        if dt not in ('f32', 'bf16'):
            raise ValueError(f"args.gemm_dtype must be 'f32' or 'bf16', got {dt!r}")
        self.bf16 = dt == 'bf16'
        # mesh_blend (round 5), fp32 builds only -- gemm_dtype='bf16' blends in plain bf16 whatever it says:
        #   'f32_split' (default): the fused mesh term's pose blend (76 % of that kernel's matrix-pipe cycles) and vertex->joint adjoint
        #       on the 16-bit matrix cores in fp32-EQUIVALENT arithmetic -- two fp16 pieces per operand (11 + 11 bits + the remainder's
        #       sign: the fp32 value to one ulp), three exact piece products per fp32 product, fp32 accumulation (nemo_v2v_fused_split).
        #       Its error against a float64 evaluation equals the fp32-MFMA kernel's
        #       (tests/test_gpu_ops.py::test_v2v_fused_split_is_fp32_equivalent: dA 1.567e-7 against 1.578e-7 rms), every 1e-4
        #       parity gate passes unchanged, and the launch is 39 % shorter (316 against 519 us at 8 x 300).
        #   'f32': the products on the fp32 MFMA pipe (rounds 1 - 4; bench.py's `f32_mfma_blend` leg).
        # args.mesh_blend, overridden by NEMO_MESH_BLEND.
        mb = os.environ.get('NEMO_MESH_BLEND') or getattr(args, 'mesh_blend', None) or 'f32_split'
        if mb not in ('f32', 'f32_split'):
            raise ValueError(f"args.mesh_blend must be 'f32' or 'f32_split', got {mb!r}")
        self.mesh_split = mb == 'f32_split' and not self.bf16
        # mlp_gemm (round 6), fp32 builds only -- the nn.Linear products of MotionNet (forward, activation and parameter gradients;
        # nemo/neural_motion_model.py:58-71, :130-148) on the 16-bit matrix cores in fp32-EQUIVALENT arithmetic (csrc/gemm_xp.h,
        # nemo_gemm_xp), from XP_MIN_ROWS rows on (below, the fp32 kernels' smaller tiles win):
        #   'f32_split' (default) = 'f32_split2': two fp16 pieces per operand (11 + 11 bits + the remainder's sign), three piece products,
        #       fp32 accumulation; a k-block of a row is exactly one 128-byte line.  fp16's narrow exponent is handled ON THE DEVICE: every
        #       copy carries a scale record {power-of-two scale, absmax}; cast-made copies are scaled from their absmax, GEMM-made copies
        #       from the bound |alpha| K absmax_A absmax_B + absmax_bias on their entries -- no copy can overflow, whatever the magnitudes
        #       (include/nemo_hip.h; tests/test_gpu_xp.py::test_fp16_pieces_cannot_overflow_and_degrade_gracefully).
        #   'f32_split3': three bf16 pieces (8 + 8 + 8 bits = the fp32 value exactly, fp32's exponent range: no scales at all), six piece
        #       products -- 1.5 x the bytes, 2 x the matrix work (8 x 300: 1.096 against 1.089 ms, C4: 81.9 against 71.8 ms).
        #   'f32': v_mfma_f32_32x32x2_f32 throughout (rounds 1 - 5; bench.py's `f32_mfma` leg; C4 92.4 ms).
        # Error against float64 <= the fp32-MFMA GEMM's in every role (tests/test_gpu_xp.py).  args.mlp_gemm, overridden by NEMO_MLP_GEMM.
        mg = os.environ.get('NEMO_MLP_GEMM') or getattr(args, 'mlp_gemm', None) or 'f32_split'
        if mg not in ('f32', 'f32_split', 'f32_split2', 'f32_split3'):
            raise ValueError(f"args.mlp_gemm must be 'f32', 'f32_split' (= 'f32_split2') or 'f32_split3', got {mg!r}")
        self.mlp_split = mg != 'f32' and not self.bf16 and version >= 1
        self.xp_fmt = 3 if mg == 'f32_split3' else 2
        # bf16 operands IN MEMORY (round 3): every dense product of the MotionNet / VPoser chain reads bf16 copies of its
        # operands (written by the producing GEMM's epilogue, plain and transposed, or by nemo_cast_bf16) through
        # nemo_gemm_bf16mem -- the same values enter the matrix cores as with the on-the-fly rounding of nemo_gemm_bf16
        # (which NemoV0's three networks keep), half the bytes move
        self.b16mem = self.bf16 and version >= 1
        self.din = (self.D if self.D > 0 else 1) + self.C
        self.ldx = (self.din + 3) // 4 * 4          # row stride of the MLP input / its gradient (16-byte rows)
        self.cx, self.cy = float(img_d0 // 2), float(img_d1 // 2)       # :3104-3106 (sic)
        self.kernel_id = RBF_KERNELS[args.rbf_kernel] if self.D > 0 else 0
        # 25 output joints out of the 49-joint map (:3662 / :3997)
        jm = [int(x) for x in assets['joint_map']]
        idx = list(range(0, 25)) if version == 4 else [38] + list(range(1, 25))
        self.ctx = SmplContext(assets, [jm[i] for i in idx], self.device)
        self.ctx49 = None      # lazily: all 49 joints (get_preds API)
        self._assets = assets
        self._jm = jm
        self.NV = self.ctx.NV
        self.vp = fold_vposer(vposer_sd, self.device)
        # mesh_blend 'f32_split': the blend-shape ADJOINT dPF = dVP P^T in the same arithmetic from SPLIT_ADJ_ROWS samples on -- the
        # mesh kernel hands d vp over as two fp16 piece planes (nemo_v2v_fused_splitmem), the blend shapes' two planes are made
        # here, the product runs on the 64 x 208 tile with fp16 MFMAs (nemo_gemm_f16x2mem_adj); NEMO_SPLIT_ADJOINT=0: fp32 GEMM
        self.split_adj = self.mesh_split and os.environ.get('NEMO_SPLIT_ADJOINT', '1') != '0'
        if self.split_adj:
            Pf = torch.as_tensor(np.asarray(assets['posedirs']), dtype=torch.float32).reshape(207, -1)
            pmax = float(Pf.abs().max())
            self.ph_scale = 2.0 ** (14 - math.frexp(pmax)[1]) if pmax > 0 else 1.0          # max |P| * scale in [2^13, 2^14)
            x = Pf * self.ph_scale
            h0 = x.half()
            h1 = (x - h0.float()).half()
            self.Ph = torch.zeros(2, 207, self.ctx.ldP, dtype=torch.int16, device=self.device)
            self.Ph[0, :, :Pf.shape[1]] = h0.view(torch.int16).to(self.device)
            self.Ph[1, :, :Pf.shape[1]] = h1.view(torch.int16).to(self.device)
        # Round 6: the split-precision adjoint through nemo_gemm_xp (fmt 2) -- d vp as ONE xp matrix (nemo_v2v_fused_splitxp), the blend
        # shapes' xp copy made here (scaled from their absmax on the device): one pass over both operands per tile with all three piece
        # products (the plane-pair form above streams A0 / B0 twice): 150 -> ~90 us per 8 x 300 launch.  NEMO_ADJ_XP=0: the round-5 pair.
        self.adj_xp = self.split_adj and os.environ.get('NEMO_ADJ_XP', '1') != '0'
        if self.adj_xp:
            self.Px = torch.zeros(207, int(self.lib.nemo_xp_ld(2, 3 * self.NV)), dtype=torch.int16, device=self.device)
            self.adj_meta = torch.zeros(2, 64, dtype=torch.float32, device=self.device)     # records of [d vp (scale 2^12), blend shapes]
            self.adj_meta[0, 0] = 4096.0
            am = (_lib.AbsmaxDesc * 1)()
            am[0].src, am[0].rows, am[0].cols, am[0].lds, am[0].meta, am[0].overwrite = (self.ctx.posedirs, 207, 3 * self.NV, self.ctx.ldP,
                                                                                          self.adj_meta[1].data_ptr(), 1)
            check(self.lib.nemo_absmax_multi(1, am, _stream()), 'nemo_absmax_multi')
            cd = (_lib.CastXpDesc * 1)()
            cd[0].src, cd[0].rows, cd[0].cols, cd[0].lds = self.ctx.posedirs, 207, 3 * self.NV, self.ctx.ldP
            cd[0].dst, cd[0].ldd, cd[0].dstT, cd[0].lddT, cd[0].scale, cd[0].meta = self.Px.data_ptr(), self.Px.stride(0), None, 0, 1.0, self.adj_meta[1].data_ptr()
            check(self.lib.nemo_cast_xp(2, 1, cd, _stream()), 'nemo_cast_xp')
        if self.b16mem:
            r8 = lambda n: (n + 7) // 8 * 8
            # blend shapes as a plain bf16 [207][3 NVp] matrix: the k-contiguous B operand of the adjoint product
            self.Pb = torch.zeros(207, self.ctx.ldP, dtype=torch.int16, device=self.device)
            self._cast(207, 3 * self.NV, self.ctx.posedirs, self.ctx.ldP, self.Pb, 0)
            self.vpb = {}
            for k in ('e2w', 'emw', 'd0w', 'd0mw', 'd3w', 'd5w'):       # (out, in) -> bf16 [out][in padded]
                wt = self.vp[k]
                self.vpb[k] = torch.zeros(wt.shape[0], r8(wt.shape[1]), dtype=torch.int16, device=self.device)
                self._cast(wt.shape[0], wt.shape[1], wt.data_ptr(), wt.stride(0), self.vpb[k], 0)
            for k in ('emw', 'e2w'):                                      # transposes for the KL backward: [in][out padded]
                wt = self.vp[k]
                self.vpb[k + 'T'] = torch.zeros(wt.shape[1], r8(wt.shape[0]), dtype=torch.int16, device=self.device)
                self._cast(wt.shape[0], wt.shape[1], wt.data_ptr(), wt.stride(0), self.vpb[k + 'T'], 1)
        self.gmm = gmm_constants(gmm, self.device)
        f32 = dict(dtype=torch.float32, device=self.device)
        self.targets = targets.to(**f32).contiguous()
        self.gt_size = gt_size.to(**f32).contiguous()
        self.hmr_theta = hmr_theta.to(**f32).contiguous()
        self.hmr_mask = hmr_mask.to(**f32).contiguous()
        # flat parameter / gradient / Adam-moment buffers
        self.layout = ParamLayout(V, self.K, self.D, self.C, self.h, self.din, version)
        n = self.layout.total
        # row stride between the phase networks of consecutive views inside the flat buffer (NOT 2K: every tensor
        # starts on a 16-byte boundary, so K % 4 != 0 leaves pads between them)
        ent = self.layout.entries
        self.ldp = (ent['phase_networks.1.shifts'][0] - ent['phase_networks.0.shifts'][0]) if V > 1 else 2 * self.K + 8
        self.params = torch.zeros(n, **f32)
        self.grads = torch.zeros(n, **f32)
        self.exp_avg = torch.zeros(n, **f32)
        self.exp_avg_sq = torch.zeros(n, **f32)
        self.betas = torch.zeros(1, 10, **f32)
        self.scal = torch.zeros(8, **f32)
        self._scal_host = torch.zeros(8, dtype=torch.float32).pin_memory()
        # early loss read-back (publish_scalars / wait_scalars): 8 values + a flag in pinned memory
        self._pub_host = torch.zeros(16, dtype=torch.float32).pin_memory()
        self._pub_np = self._pub_host.numpy()
        self._pub_flag = self._pub_np.view(np.int32)[8:9]
        self.early_readback = os.environ.get('NEMO_EARLY_READBACK', '1') != '0'
        self.ws = {}
        self.side_stream = torch.cuda.Stream(device=self.device)   # prior branches of the step (see _forward_backward)
        self.side_stream2 = torch.cuda.Stream(device=self.device)
        self.comm_stream = torch.cuda.Stream(device=self.device)    # bucketed gradient all-reduces + their Adam (dist.py)
        # arena of the ordered (deterministic) reductions: owned by THIS engine, on its device, bound to the calling thread at the top
        # of every pass (include/nemo_hip.h nemo_reduce_ws_bind, ABI 17): engines on other devices / streams have their own
        self.red_ws = torch.zeros(int(self.lib.nemo_reduce_ws_bytes(max(V * T, 8192), V)) // 4, **f32)
        # split-K scratch of nemo_gemm_f32 (arrival tickets + partial tiles), one per stream that launches GEMMs
        self.gemm_ws = [torch.zeros(16 << 20, device=self.device) for _ in range(3)]
        self._colsums = []
        self._seg_host = self._seg_dev = self._seg_pending = None
        self.timers = None
        self.detach_articulation = False
        self.start_global_traj_anywhere = False
        # device scalar 'number of real samples' of a PADDED step (include/nemo_hip.h, nemo_kp_fwd), else None; set by
        # MultiViewModel.step around its launches only
        self.nvalid = None
        # device int64[V]: samples per view of the batch being launched (known from the indices: MultiViewModel._stage_indices),
        # or None -- then the key-point objective runs as two launches (nemo_kp_fwd, nemo_kp_bwd_ex)
        self.view_cnt = None

    def bind_reduce_ws(self):
        """Bind this engine's reduction arena to the calling host thread and rewind it (top of every pass)."""
        check(self.lib.nemo_reduce_ws_bind(self.red_ws.data_ptr(), self.red_ws.numel() * 4), 'nemo_reduce_ws_bind')

    # ------------------------------------------------------------------ parameter views
    def view(self, name, buf=None):
        off, shape = self.layout.entries[name]
        buf = self.params if buf is None else buf
        return buf[off:off + int(np.prod(shape))].view(shape)

    def p(self, name):
        return self.view(name).data_ptr()

    def g(self, name):
        return self.view(name, self.grads).data_ptr()

    # ------------------------------------------------------------------ workspaces
    DW_ASIDE_ROWS = 65536           # fp32 chain: hidden-layer parameter-gradient products on the side stream up to this many rows
    # bf16 chain: the first layer (K = 105) and its two backward products on the chain's bf16 kernels too (round 5; False: fp32
    # products over nn.Linear(105, h)'s unaligned rows, rounds 3 - 4); NEMO_B16_FIRST_LAYER=0 restores that
    B16_FIRST_LAYER = os.environ.get('NEMO_B16_FIRST_LAYER', '1') != '0'
    # the batch of the pass in flight is sorted by view (the model sets it for full batches: view-major by construction); lets
    # the phase backward find a view's samples by a search instead of a scan.  Part of a captured launch: full and random
    # batches have different graph keys already.
    batch_sorted = False
    B16_DW_ASIDE_ROWS = 10000       # bf16 chain: parameter-gradient products on the side stream up to this many rows
    SMALL_BATCH_ROWS = 1024  # backward_mlp: below this many rows the dW GEMMs run beside the dX chain
    GROUPED_DW_ROWS = 400    # ... and up to this many as ONE grouped launch behind it (a one-instance shard)
    SPLIT_ADJ_ROWS = 256     # mesh_blend 'f32_split': the blend-shape adjoint in split precision from this many samples on
    XP_MIN_ROWS = int(os.environ.get('NEMO_XP_MIN_ROWS', '1600'))    # mlp_gemm 'f32_split': the chain on nemo_gemm_xp from this many rows on
    XP_DW_ASIDE_ROWS = 65536  # ... its parameter-gradient products on the side stream up to this many rows
    XP_GROUPED_DW_ROWS = int(os.environ.get('NEMO_XP_GROUPED_DW_ROWS', '8192'))   # ... as ONE grouped launch up to this many rows
    # ... and the frozen VPoser's products (encode, decode, the KL term's adjoint) from this many rows on.  OFF by default (measured, same
    # box, xp from 8192 rows / fp32: 40 x 300 3.39 - 3.45 / 3.38 - 3.40 ms, C4 68.4 - 69.7 / 68.6 - 68.7 ms): the fp32 launches are 15 of
    # C4's 87 ms of kernel time but run BESIDE the key-point and prior branches, and a nemo_gemm_xp launch takes whole CUs (160 KB of
    # LDS) -- what its 3x shorter launches save, the branches beside them lose.  NEMO_VP_XP_MIN_ROWS=<rows> switches it on.
    VP_XP_MIN_ROWS = int(os.environ.get('NEMO_VP_XP_MIN_ROWS', str(1 << 62)))
    XMETA = ('X', 'H1', 'H2', 'H3', 'dHEAD', 'dH', 'dH_b', 'dH_c', 'W0', 'W2', 'W4', 'Whead', 'b0', 'b2', 'b4',
             'vAA', 'vE1', 'vD1', 'vD2', 'vdM', 'vdE')
    VPMETA = ('e2w', 'd0mw', 'd3w', 'd5w', 'emw', 'e2b', 'd0mb', 'd3b')
    # (fp32 backward_mlp: the bias gradients come from the dX launches' per-band column sums at EVERY size -- headline 1.257 / 1.259 ms
    #  without / with, C4 101.8 / 101.5 ms: the win is C4's 6.6 GB of reads; a row threshold was never applied and is gone)
    MAX_WORKSPACES = 24      # distinct batch sizes kept alive (a rank of a sharded minibatch run sees many)

    @staticmethod
    def ws_capacity(N):
        """Workspaces are shared by batch sizes: below 2048 samples a size is rounded up to a multiple of 64 (one rank's
        share of a random minibatch takes 60 - 80 different values around B / world: a workspace -- ~50 allocations --
        per value would be evicted and rebuilt all the time).  Every kernel takes the actual N; only the captured
        graphs are per N (they are keyed by it inside the shared workspace)."""
        return N if N >= 2048 else max(64, (N + 63) // 64 * 64)

    def _ws(self, n_samples):
        N = self.ws_capacity(n_samples)
        w = self.ws.get(N)
        if w is not None:
            self.ws[N] = self.ws.pop(N)          # most recently used last
            return w
        while len(self.ws) >= self.MAX_WORKSPACES:
            self.ws.pop(next(iter(self.ws)))     # drop the least recently used workspace (and its graphs)
        f32 = dict(dtype=torch.float32, device=self.device)
        Z = lambda *s: torch.zeros(*s, **f32)
        h, nq = self.h, self.ctx.nq
        Nc = min(N, 8192)
        # everything that must be zero at the start of a step lives in ONE arena (a single memset)
        # (incl. this workspace's loss-scalar slots: FitEngine.scal points at the active workspace's)
        # (xmeta: the scale records of the split-precision MLP chain's copies, include/nemo_hip.h nemo_gemm_xp fmt 2: absmax
        #  accumulators must be zero at the start of a step)
        sizes = OrderedDict(scal=(8,), xmeta=(len(self.XMETA), 64), view_acc=(self.V, 2), dAA=(N, 72), dJp=(N, 24, 3), dA2=(Nc, 24, 12),
                            dPF2=(Nc, 208))
        arena = Z(sum(int(np.prod(v)) for v in sizes.values()))
        views, off = {}, 0
        for k, shp in sizes.items():
            n_ = int(np.prod(shp))
            views[k] = arena[off:off + n_].view(shp)
            off += n_
        w = dict(
            X=Z(N + 1, self.ldx), H1=Z(N + 1, h), H2=Z(N + 1, h), H3=Z(N + 1, h), HEAD=Z(N + 1, HEAD_LD),
            phase=Z(N), phase_ws=Z(N), gmm_ws=Z(N, self.gmm['M']), R=Z(N, 24, 9), AA=Z(N, 72), A=Z(N, 24, 12), Jp=Z(N, 24, 3),
            PF=Z(N, 208), Mq=Z(N, max(nq * 72, 1)), j3d=Z(N, self.ctx.n_out, 3), p2d=Z(N, self.ctx.n_out, 2),
            loss_all=Z(N, self.ctx.n_out, 2), norm=Z(1), dj3d=Z(N, self.ctx.n_out, 3),
            E1=Z(N, 512), MULV=Z(N, 64), D1=Z(N, 512), D2=Z(N, 512),
            D3=Z(N, 126), AAdec=Z(N, 63),
            R2=Z(2 * Nc, 24, 9), A2=Z(2 * Nc, 24, 12), Jp2=Z(2 * Nc, 24, 3), PF2=Z(2 * Nc, 208),
            # (kept at full size beside dVPh: evaluation passes, chunk tails below SPLIT_ADJ_ROWS and a model the range guard sends back
            #  to fp32 planes -- nemo_ctx_split_ok after a set_betas -- all write it; 680 MB per 8192-sample workspace of 288 GB)
            dVPt=Z(3 * self.ctx.NVp, (Nc + 15) // 16 * 16) if not self.b16mem else Z(16),
            # (mesh_blend 'f32_split': d vp as two fp16 piece planes [2][samples][blend-shape row stride])
            dVPh=(torch.zeros(2, (Nc + 15) // 16 * 16, self.ctx.ldP, dtype=torch.int16, device=self.device)
                  if (getattr(self, 'split_adj', False) and not getattr(self, 'adj_xp', False)) else None),
            # (round 6: ... or as ONE xp matrix, the pieces interleaved per k-block of 32: the adjoint through nemo_gemm_xp)
            dVPx=(torch.zeros((Nc + 15) // 16 * 16, int(self.lib.nemo_xp_ld(2, 3 * self.ctx.NVp)), dtype=torch.int16, device=self.device)
                  if getattr(self, 'adj_xp', False) else None),
            dR2=Z(N, 24, 9),
            dR=Z(N, 24, 9), dA=Z(N, 24, 12), dMq=Z(N, max(nq * 72, 1)),
            dPF=Z(N, 208), dHEAD=Z(N + 1, HEAD_LD), dH=Z(N + 1, h), dH_b=Z(N + 1, h), dH_c=Z(N + 1, h),
            dX=Z(N + 1, self.ldx), dMULV=Z(N, 64), dE_a=Z(N, 512), Nc=Nc,
            # device-resident step inputs of a captured graph: the indices, the number of real samples of a padded step
            # (element N of vi_static) and, behind it, the V per-view sample counts of the batch (nemo_kp_fwd_bwd)
            vi_static=torch.zeros(N + 1 + self.V, dtype=torch.long, device=self.device),
            fi_static=torch.zeros(N + 1, dtype=torch.long, device=self.device), graphs={}, cap=N)
        w.update(views)
        w['zero_arena'] = arena
        if self.b16mem:
            # bf16 copies (plain: [row][feature], k-contiguous for the forward / dX products; T: [feature][row] for the
            # parameter gradients) of every operand of the dense chain; zero-initialised: pad columns are never written
            i16 = dict(dtype=torch.int16, device=self.device)
            r8 = lambda n: (n + 7) // 8 * 8
            rp, hp = r8(N + 1), r8(h)
            Zb = lambda *sh: torch.zeros(*sh, **i16)
            w.update(dVPb=Zb((Nc + 15) // 16 * 16, self.ctx.ldP), dHEADb=Zb(N + 1, 152), dHEADbT=Zb(147, rp),
                     AAb=Zb(N, 64), E1b=Zb(N, 512), MULVb=Zb(N, 64), D1b=Zb(N, 512), D2b=Zb(N, 512), dMULVb=Zb(N, 64),
                     dE_ab=Zb(N, 512))
            for k in ('H1', 'H2', 'H3', 'dH', 'dH_b', 'dH_c'):
                w[k + 'b'], w[k + 'bT'] = Zb(N + 1, hp), Zb(h, rp)
            # the network input (RBF features + code) as bf16, plain and transposed: first layer on the bf16 chain (round 5)
            # (forward: three bf16 pieces per row -- [hi | lo | hi] against the weight's [hi | hi | lo]: the first layer's product in
            #  fp32-equivalent split precision; backward: the plain transposed copy)
            w['Xs'], w['XbT'] = Zb(N + 1, 3 * r8(self.din)), Zb(self.din, rp)
        if self.mlp_split and N + 1 >= self.XP_MIN_ROWS:
            # xp copies (csrc/gemm_xp.h: three bf16 pieces per value, k-blocks of 32) of every operand of the MotionNet chain: plain
            # [row][feature] for the forward / dX products, T [feature][row] for the parameter gradients; zero-initialised
            i16 = dict(dtype=torch.int16, device=self.device)
            xl = lambda k: int(self.lib.nemo_xp_ld(self.xp_fmt, k))
            Zx = lambda rows, k: torch.zeros(rows, xl(k), **i16)
            w.update(Xx=Zx(N + 1, self.din), XxT=Zx(self.din, N + 1), dHEADx=Zx(N + 1, 147), dHEADxT=Zx(147, N + 1))
            # X's scale record: OUTSIDE the zero-filled arena (the phase kernel accumulates into it in the same launch that zero-fills
            # the arena); its absmax slots are returned to zero by the forward's last product (nemo_gemm_xp metaZero)
            w['xmeta_x'] = torch.zeros(64, dtype=torch.float32, device=self.device)
            for k in ('H1', 'H2', 'H3', 'dH', 'dH_b', 'dH_c'):
                w[k + 'x'], w[k + 'xT'] = Zx(N + 1, h), Zx(h, N + 1)
            if N >= self.VP_XP_MIN_ROWS and self.version >= 1:
                # ... and of the VPoser chain's operands (forward_vposer / vposer_mulv / backward_vposer_kl); the frozen weights' copies
                # and records are made here, once, outside any capture
                w.update(AAx=Zx(N, 63), E1x=Zx(N, 512), D1x=Zx(N, 512), D2x=Zx(N, 512), dMULVx=Zx(N, 64), dE_ax=Zx(N, 512))
                self._vposer_xp_weights()
        # per-band column sums of the three activation gradients of the MotionNet backward (the epilogues of their launches:
        # nemo_gemm_bf16mem(colsum) / nemo_gemm_f32_colsum): the bias gradients of layers 4, 2, 0 from 2 ceil(r / 64) short rows
        R = int(self.lib.nemo_gemm_colsum_rows(N + 1))
        w['cs4'], w['cs2'], w['cs0'] = Z(R, h), Z(R, h), Z(R, h)
        if self.version == 0:            # hidden activations of the orient and translation networks (poses: H1..H3)
            w.update(O1=Z(N + 1, h), O2=Z(N + 1, h), O3=Z(N + 1, h), T1=Z(N + 1, h), T2=Z(N + 1, h))
        # scratch of nemo_v2v_fused (arrival tickets, zero at allocation and returned to zero by the kernel, +
        # per-vertex-range partial dA): owned by THIS workspace and sized for every chunk length it launches, so a
        # HIP graph captured over the workspace never sees the buffer replaced under it
        # (every launch size this workspace can see: whole chunks + the ragged rest of an exact large batch, or any
        #  number of 16-sample groups up to the capacity of a shared small one)
        tail = N - (N // Nc) * Nc
        sizes_seen = {Nc, tail or Nc} if N >= 2048 else set(range(16, N + 1, 16))
        need = max(int(self.lib.nemo_v2v_fused_ws_bytes(self.ctx.handle, n_)) for n_ in sizes_seen)
        w['mesh_ws'] = torch.zeros((need + 3) // 4, dtype=torch.float32, device=self.device)
        # strided views into the merged MLP-head buffers
        w['ROT'], w['TR'] = w['HEAD'][:, :144], w['HEAD'][:, 144:147]
        w['dROT'], w['dTR'] = w['dHEAD'][:, :144], w['dHEAD'][:, 144:147]
        self.ws[N] = w
        return w

    # ------------------------------------------------------------------ kernel helpers
    def gemm(self, ta, tb, M, N, K, A, lda, B, ldb, Cp, ldc, bias=None, act=0, mask=None, ldmask=0,
             mask_mode=0, alpha=1.0, out_mode=0, split_k=0, tag=None, dense=False, colsum=None):
        """split_k 0: the library picks the tile shape and the K split (combined inside the launch through
        this stream's scratch).  dense=True marks the contractions that ``args.gemm_dtype = 'bf16'`` moves to the
        bf16 matrix cores (nn.Linear forward / backward of MotionNet and VPoser, the blend-shape adjoint); the joint
        functionals (PF @ C1, millimetre-sensitive) always stay fp32."""
        ev = self._event_begin(tag, 2.0 * M * N * K)
        cur = torch.cuda.current_stream()
        ws = self.gemm_ws[1 if cur == self.side_stream else (2 if cur == self.side_stream2 else 0)]
        if colsum is not None:
            # fp32 only: the launch also leaves the per-band column sums of its result (nemo_gemm_f32_colsum, round 5)
            assert not self.bf16 and out_mode == 0 and split_k == 0
            check(self.lib.nemo_gemm_f32_colsum(ta, tb, M, N, K, A, lda, B, ldb, Cp, ldc, bias, act, mask, ldmask, mask_mode, alpha,
                                                colsum.data_ptr(), colsum.stride(0), ws.data_ptr(), ws.numel() * 4, _stream()),
                  'nemo_gemm_f32_colsum')
            self._event_end(ev)
            return
        fn = self.lib.nemo_gemm_bf16 if (dense and self.bf16) else self.lib.nemo_gemm_f32
        check(fn(ta, tb, M, N, K, A, lda, B, ldb, Cp, ldc, bias, act, mask, ldmask, mask_mode, alpha, out_mode,
                 split_k, ws.data_ptr(), ws.numel() * 4, _stream()), 'nemo_gemm')
        self._event_end(ev)

    def kernel_flops_by_pipe(self, tag, flops):
        """Split a tagged (bench-timed) kernel's algorithmic FLOPs by the matrix pipe they run on, so that bench.py can
        price each part against its own peak.  fp32 build: everything on the fp32 MFMA pipe.  gemm_dtype='bf16': the
        tagged GEMMs run on the bf16 pipe; of the fused mesh kernel the two pose blends and the vertex->joint adjoint do
        (2 x 3 x 207 + 288 of its 2 x 3 x 207 + 2 x 288 + 288 multiply-adds per vertex and sample) -- skinning and L1 stay
        fp32."""
        # (sparse skinning -- <= 4 non-zero weights per vertex -- is fp32 FMA work on the VALU, not on a matrix pipe: its own
        #  entry, priced at the fp32 vector peak, which equals the fp32 MFMA peak on this part)
        valu = 0.0
        b16_skin = self.bf16 and os.environ.get('NEMO_MESH_SPLIT', '2') != '3'      # kernel MODE 2: skinning as split-precision MFMAs
        skin16 = 0.0
        if tag == 'mesh_v2v_fused' and self._mesh_mode6():
            # (kernel MODE 6: the two skinnings on the 16-bit pipe as FOUR fp16 piece products per algorithmic product -- pipe 'f16x4';
            #  the algorithmic work stays what mesh_macs counts: the model's non-zero weights)
            skin16 = flops * (2 * 12 * (4 if self.ctx.skin_sparse_flag else 24)) / self.mesh_macs()
        elif tag == 'mesh_v2v_fused' and self.ctx.skin_sparse_flag and not b16_skin:
            valu = flops * (2 * 12 * 4) / self.mesh_macs()
        if not self.bf16:
            if tag in ('gemm_mlp_hidden_fwd', 'gemm_mlp_hidden_dx') and self.mlp_split and any('Xx' in w_ for w_ in self.ws.values()):
                return {('f16x3' if self.xp_fmt == 2 else 'bf16x6'): flops}      # (nemo_gemm_xp: three fp16 / six bf16 piece products per product)
            if tag == 'gemm_pose_blend_bwd' and self.split_adj and self.ctx.split_ok:
                return {'f16x3': flops}              # (nemo_gemm_f16x2mem_adj: three fp16 piece products per algorithmic product)
            out = {'f32': flops - valu}
            if tag == 'mesh_v2v_fused' and self.mesh_split:
                # (csrc/smpl.hip MODE 5: the two pose blends and the vertex->joint adjoint run on the 16-bit matrix pipe as THREE fp16
                #  piece products per algorithmic product -- pipe 'f16x3', whose peak is a third of the fp16 / bf16 MFMA peak;
                #  NEMO_MESH_PIECES=3 (MODE 4): six bf16 piece products, 'bf16x6')
                b16 = flops * (2 * 3 * 207 + 288) / self.mesh_macs()
                out = {('bf16x6' if self._mesh_three_pieces() else 'f16x3'): b16, 'f32': flops - b16 - valu - skin16}
                if skin16:
                    out['f16x4'] = skin16
            if valu:
                out['valu_f32'] = valu
            return out
        if tag == 'mesh_v2v_fused':
            # (csrc/smpl.hip MODE 3: the vertex->joint adjoint -- 288 of the multiply-adds -- runs on the bf16 pipe too, as
            #  four bf16 piece products per algorithmic product)
            on16 = 2 * 3 * 207 + 288
            if b16_skin:        # (MODE 2, the default: the two skinnings too -- their algorithmic work, the kernel executes the dense product)
                on16 = self.mesh_macs()
            b16 = flops * on16 / self.mesh_macs()
            out = {'bf16': b16, 'f32': flops - b16 - valu}
            if valu:
                out['valu_f32'] = valu
            return out
        return {'bf16': flops}

    def mesh_macs(self, strict=False):
        """Multiply-adds of the fused mesh kernel per (vertex, sample): two pose blends (3 x 207 each), two skinnings and
        the vertex->joint adjoint (24 x 12).  A skinning is 24 x 12 as the dense product of lbs.py:236-241, or
        skin_nnz x 12 on the VALU when the model's weights are sparse (<= 4 non-zero per vertex: csrc/smpl.hip SPARSE).
        ``strict``: the adjoint dA = W^T dT priced at the weight matrix's real sparsity too (skin_nnz x 12 instead of the
        24 x 12 the kernel's two joint tiles execute) -- bench.py's `frac_strict`."""
        sparse = self.ctx.skin_sparse_flag
        skin = 12 * (4 if sparse else 24)
        return 2 * 3 * 207 + 2 * skin + (skin if strict and sparse else 288)

    def _mesh_three_pieces(self):
        """mesh_blend 'f32_split' runs its three-bf16-piece form (kernel MODE 4): asked for (NEMO_MESH_PIECES=3, without fp16 planes of
        d vp) or forced by the range guard (nemo_ctx_split_ok == 0)."""
        return (os.environ.get('NEMO_MESH_PIECES') == '3' and not self.split_adj) or not self.ctx.split_ok

    def mesh_blend_in_effect(self):
        """What bench.py reports as config.mesh_blend."""
        if self.bf16:
            return 'bf16'
        if not self.mesh_split:
            return 'f32'
        return 'f32_split3 (range guard: vp bound %.3g)' % self.ctx.vp_bound if not self.ctx.split_ok else ('f32_split3' if self._mesh_three_pieces() else 'f32_split')

    def mesh_kernel_variant(self):
        """The mesh kernel instantiation this engine launches (what counter files under profiles/ are keyed by)."""
        mode = (3 if os.environ.get('NEMO_MESH_SPLIT', '2') == '3' else 2) if self.bf16 else ((4 if self._mesh_three_pieces() else (6 if self.ctx.skin_mfma else 5)) if self.mesh_split else 0)
        return f"mesh_v2v_fused_kernel<{mode}, {'true' if self.ctx.skin_sparse_flag and mode not in (2, 6) else 'false'}>"

    def _mesh_mode6(self):
        return (not self.bf16) and self.mesh_split and not self._mesh_three_pieces() and self.ctx.skin_mfma

    # Optional per-launch HIP-event timing of tagged kernels (bench.py's roofline leg).  Events are
    # recorded on the stream the kernels are launched on (torch's current stream).
    def _event_begin(self, tag, flops):
        if self.timers is None or tag is None:
            return None
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        self.timers.setdefault(tag, []).append((a, b, flops))
        return b

    @staticmethod
    def _event_end(ev):
        if ev is not None:
            ev.record()

    def _linear(self, rows, x, ldx, fin, w, b, fout, y, ldy, act=0, tag=None, ldw=None):
        """y = act(x @ w^T + b), w stored (fout, fin) like nn.Linear (row stride ldw, default fin)."""
        self.gemm(0, 1, rows, fout, fin, x, ldx, w, fin if ldw is None else ldw, y, ldy, bias=b, act=act, tag=tag,
                  dense=True)

    def _linear_bwd_params(self, rows, x, ldx, fin, dy, lddy, fout, gw, gb, nbias=None):
        """gw (fout,fin) += dy^T @ x ;  gb[:nbias] += colsum(dy) (skipped when gb is None).  K = rows
        is split for occupancy."""
        self.gemm(1, 0, fout, fin, rows, dy, lddy, x, ldx, gw, fin, out_mode=1, dense=True)
        if gb is not None:       # bias gradients are batched into one launch (flush_colsums)
            self._colsums.append((dy, rows, fout if nbias is None else nbias, lddy, gb))

    # ---- bf16-in-memory chain (self.b16mem) -----------------------------------------------------------------
    def _cast(self, rows, cols, src_ptr, lds, dst, transpose):
        check(self.lib.nemo_cast_bf16(rows, cols, src_ptr, lds, dst.data_ptr(), dst.stride(0), transpose, _stream()),
              'nemo_cast_bf16')

    def gemm16(self, M, N, K, Ab, Bb, Cp, ldc, bias=None, act=0, mask=None, ldmask=0, mask_mode=0, alpha=1.0, out_mode=0,
               Cb=None, CbT=None, colsum=None, tag=None):
        """C (M x N, fp32) (op)= epilogue(alpha * A B^T) with A (M x K), B (N x K) bf16 in memory (2-D int16 tensors whose
        row stride is the ld; K is taken up to the next even number: the pad column is zero by construction)."""
        ev = self._event_begin(tag, 2.0 * M * N * K)
        cur = torch.cuda.current_stream()
        ws = self.gemm_ws[1 if cur == self.side_stream else (2 if cur == self.side_stream2 else 0)]
        check(self.lib.nemo_gemm_bf16mem(M, N, (K + 1) // 2 * 2, Ab.data_ptr(), Ab.stride(0), Bb.data_ptr(), Bb.stride(0), Cp, ldc,
                                         bias, act, mask, ldmask, mask_mode, alpha, out_mode,
                                         dptr(Cb), Cb.stride(0) if Cb is not None else 0,
                                         dptr(CbT), CbT.stride(0) if CbT is not None else 0,
                                         dptr(colsum), colsum.stride(0) if colsum is not None else 0,
                                         ws.data_ptr(), ws.numel() * 4, _stream()), 'nemo_gemm_bf16mem')
        self._event_end(ev)

    def _weights_b16(self, transposed):
        """bf16 copies of the four MotionNet weight matrices for this step (they change in Adam at the step's end): plain
        [out][in] for the forward, transposed [in][out] for the activation gradients."""
        lm, h = 'learned_motion.', self.h
        if not hasattr(self, '_wb'):
            i16 = dict(dtype=torch.int16, device=self.device)
            r8 = lambda n: (n + 7) // 8 * 8
            self._wb = {'0': torch.zeros(h, r8(self.din), **i16), '2': torch.zeros(h, r8(h), **i16), '4': torch.zeros(h, r8(h), **i16),
                        'head': torch.zeros(147, r8(h), **i16), '0T': torch.zeros(self.din, r8(h), **i16),
                        '2T': torch.zeros(h, r8(h), **i16), '4T': torch.zeros(h, r8(h), **i16), 'headT': torch.zeros(h, 152, **i16)}
        shapes = (('2', lm + 'net.net.2.weight', h, h), ('4', lm + 'net.net.4.weight', h, h), ('head', lm + 'rot_out.weight', 147, h))
        if self.B16_FIRST_LAYER:
            if '0s' not in self._wb:
                self._wb['0s'] = torch.zeros(h, 3 * ((self.din + 7) // 8 * 8), dtype=torch.int16, device=self.device)
            check(self.lib.nemo_cast_bf16_split3(h, self.din, self.p(lm + 'net.net.0.weight'), self.din, self._wb['0s'].data_ptr(),
                                                 self._wb['0s'].stride(0), 1, _stream()), 'nemo_cast_bf16_split3')
            if transposed:
                self._cast(h, self.din, self.p(lm + 'net.net.0.weight'), self.din, self._wb['0T'], 1)
        for key, name, fo, fi in shapes:
            for tr in ((0, 1) if transposed == 'both' else (1,) if transposed else (0,)):
                self._cast(fo, fi, self.p(name), fi, self._wb[key + ('T' if tr else '')], tr)
        return self._wb

    def _forward_nets_b16(self, w, N, train):
        """forward_pose's MLP on the bf16-in-memory chain: every layer reads the previous layer's bf16 copy and leaves its
        own (plain for the next layer, transposed for its parameter gradient when `train`)."""
        lm, h, r = 'learned_motion.', self.h, N + 1
        # the weights' bf16 copies (plain for this forward and, in a training step, the transposed ones of the backward: the
        # weights only change in Adam) are made on the side stream, beside the first layer's fp32 product
        main, side = torch.cuda.current_stream(), self.side_stream
        side.wait_event(main.record_event())
        with torch.cuda.stream(side):
            wb = self._weights_b16('both' if train else False)
            casts_done = side.record_event()
        self._wbT_fresh = bool(train)
        T = lambda k: w[k] if train else None
        # the first layer (K = 105 RBF features + code, rows of nn.Linear(105, h) not 16-byte aligned) stays where it was:
        # fp32 arithmetic (its operands never qualified for the bf16 path; 2 % of the MLP's FLOPs)
        # ... and leaves its output as bf16 straight from the epilogue (nemo_gemm_f32_b16out): H1 is only ever read as the
        # bf16 operand of the next layer, of the layer-2 parameter gradient and as a ReLU mask
        if self.B16_FIRST_LAYER:
            # Round 5: the first layer on the chain's kernels too -- at 12 001 rows the fp32 product over nn.Linear(105, h)'s
            # unaligned rows took 67 us on the step's critical path.  In SPLIT precision (three bf16 pieces per operand row,
            # K = 3 x 112: hi hi + lo hi + hi lo, the fp32 product up to 2^-17): with plainly rounded inputs the pre-activations
            # of this layer move by 2^-9, ReLU masks of the large-gradient "phase 0" row flip and single entries of the layer-0 /
            # layer-2 weight gradients are off by 30 - 70 % of the largest entry (tools/debug/bf16_first_layer_grads.py)
            check(self.lib.nemo_cast_bf16_split3(r, self.din, dptr(w['X']), self.ldx, w['Xs'].data_ptr(), w['Xs'].stride(0), 0,
                                                 _stream()), 'nemo_cast_bf16_split3')
            if train:
                self._cast(r, self.din, dptr(w['X']), self.ldx, w['XbT'], 1)
            main.wait_event(casts_done)
            self.gemm16(r, h, w['Xs'].shape[1], w['Xs'], wb['0s'], None, h, bias=self.p(lm + 'net.net.0.bias'), act=1,
                        Cb=w['H1b'], CbT=T('H1bT'))
        else:
            ws = self.gemm_ws[1 if main == self.side_stream else (2 if main == self.side_stream2 else 0)]
            check(self.lib.nemo_gemm_f32_b16out(0, 1, r, h, self.din, dptr(w['X']), self.ldx, self.p(lm + 'net.net.0.weight'),
                                                self.din, None, 0, self.p(lm + 'net.net.0.bias'), 1, w['H1b'].data_ptr(),
                                                w['H1b'].stride(0), dptr(T('H1bT')), w['H1bT'].stride(0), ws.data_ptr(),
                                                ws.numel() * 4, _stream()), 'nemo_gemm_f32_b16out')
            main.wait_event(casts_done)
        # (hidden activations exist as bf16 copies only: the ReLU masks of the backward read them -- sign and zero survive
        #  the rounding --, nothing else needs the fp32 values)
        self.gemm16(r, h, h, w['H1b'], wb['2'], None, h, bias=self.p(lm + 'net.net.2.bias'), act=1,
                    Cb=w['H2b'], CbT=T('H2bT'), tag='gemm_mlp_hidden_fwd')
        self.gemm16(r, h, h, w['H2b'], wb['4'], None, h, bias=self.p(lm + 'net.net.4.bias'), act=1,
                    Cb=w['H3b'], CbT=T('H3bT'))
        self.gemm16(r, 147, h, w['H3b'], wb['head'], dptr(w['HEAD']), HEAD_LD, bias=self.p(lm + 'rot_out.bias'))

    def _backward_mlp_b16(self, w, N, view_idx, frame_idx, raw_phase, nout, nbias, stages=(0, 1, 2), bucketed=False):
        """backward_mlp on the bf16-in-memory chain.  Every product is C = A B^T with k-contiguous bf16 operands:
        dX_l = dY_l (W_l^T)^T reads dY_l's plain copy and the transposed weight copy; dW_l = dY_l^T X_l reads the two
        TRANSPOSED activation copies (K = rows).  ``stages`` / ``bucketed`` as in backward_mlp: a bucketed (sharded) step
        runs the three stages as separate calls and wants each stage's bias column sums flushed at its end."""
        L, lm, h, r = self.lib, 'learned_motion.', self.h, N + 1
        if 0 in stages and not getattr(self, '_wbT_fresh', False):
            self._weights_b16(True)           # (no training forward ran in front of this backward: warm-up after an eval pass)
        self._wbT_fresh = False
        wb = self._wb                         # transposed copies cast in this step's forward (_forward_nets_b16) or just above
        cs = self._colsums
        # The activation gradients of the two hidden layers exist as bf16 copies only; their bias gradients come from the
        # per-band column sums the same launches leave (nemo_gemm_bf16mem(colsum)): 2 ceil(r / 64) rows instead of r
        R = int(L.nemo_gemm_colsum_rows(r))

        def end_of_stage():
            if bucketed:
                self.flush_colsums()

        # The parameter-gradient products are off the dX chain.  Un-bucketed (single GPU) they go to the side stream, each as
        # soon as its dY exists, ENQUEUED BEHIND the chain's next product (a replayed graph keeps a node's first successor on its
        # queue): the bf16 products are bound by LDS-DMA issue and latency, not by the matrix pipe (11 % of its peak), so two of
        # them side by side overlap well (the fp32 ones do too, less: backward_mlp).  Same box, ms per
        # bf16 step beside / in line: 1200 rows 0.600 / 0.659, 2400: 0.792 / 0.866, 4800: 1.298 / 1.364, 7200: 1.726 / 1.849,
        # 8100: 1.831 / 1.926, 9600: 2.257 / 2.292, 12 000 (C3): 2.68 / 2.63 -> beside up to B16_DW_ASIDE_ROWS.
        main, side = torch.cuda.current_stream(), self.side_stream
        aside = (not bucketed) and r <= self.B16_DW_ASIDE_ROWS
        pend = []

        def ready():
            return main.record_event() if aside else None

        def dW(ev, fn):
            if not aside:
                return fn()
            pend.append((ev, fn))

        def flush_dW():
            for ev, fn in pend:
                side.wait_event(ev)
                with torch.cuda.stream(side):
                    fn()
            del pend[:]

        if 0 in stages:
            self._cast(r, nout, dptr(w['dHEAD']), HEAD_LD, w['dHEADb'], 0)
            self._cast(r, nout, dptr(w['dHEAD']), HEAD_LD, w['dHEADbT'], 1)
            # heads
            ev = ready()
            dW(ev, lambda: self.gemm16(nout, h, r, w['dHEADbT'], w['H3bT'], self.g(lm + 'rot_out.weight'), h, out_mode=1))
            cs.append((dptr(w['dHEAD']), r, nbias, HEAD_LD, self.g(lm + 'rot_out.bias')))
            self.gemm16(r, h, nout, w['dHEADb'], wb['headT'], None, h, mask=dptr(w['H3b']), ldmask=w['H3b'].stride(0),
                        mask_mode=17, Cb=w['dHb'], CbT=w['dHbT'], colsum=w['cs4'])
            flush_dW()
            # layer 4
            ev = ready()
            dW(ev, lambda: self.gemm16(h, h, r, w['dHbT'], w['H2bT'], self.g(lm + 'net.net.4.weight'), h, out_mode=1))
            cs.append((dptr(w['cs4']), R, h, h, self.g(lm + 'net.net.4.bias')))
            self.gemm16(r, h, h, w['dHb'], wb['4T'], None, h, mask=dptr(w['H2b']), ldmask=w['H2b'].stride(0),
                        mask_mode=17, Cb=w['dH_bb'], CbT=w['dH_bbT'], colsum=w['cs2'], tag='gemm_mlp_hidden_dx')
            flush_dW()
            end_of_stage()
        if 1 in stages:
            # layer 2
            ev = ready()
            dW(ev, lambda: self.gemm16(h, h, r, w['dH_bbT'], w['H1bT'], self.g(lm + 'net.net.2.weight'), h, out_mode=1))
            cs.append((dptr(w['cs2']), R, h, h, self.g(lm + 'net.net.2.bias')))
            if self.B16_FIRST_LAYER:
                self.gemm16(r, h, h, w['dH_bb'], wb['2T'], None, h, mask=dptr(w['H1b']), ldmask=w['H1b'].stride(0),
                            mask_mode=17, Cb=w['dH_cb'], CbT=w['dH_cbT'], colsum=w['cs0'])
            else:
                self.gemm16(r, h, h, w['dH_bb'], wb['2T'], dptr(w['dH_c']), h, mask=dptr(w['H1b']), ldmask=w['H1b'].stride(0),
                            mask_mode=17, Cb=None, CbT=None, colsum=w['cs0'])
            flush_dW()
            end_of_stage()
        if 2 not in stages:
            return
        ev = ready()
        cs.append((dptr(w['cs0']), R, h, h, self.g(lm + 'net.net.0.bias')))
        if self.B16_FIRST_LAYER:
            # layer 0 on the chain as well: dW_0 = dY_0^T X over the two transposed copies, dX_0 = dY_0 W_0 over the plain copy
            # and the transposed weight copy (fp32 results: the weight gradient and the phase / RBF / code backward's input)
            dW(ev, lambda: self.gemm16(h, self.din, r, w['dH_cbT'], w['XbT'], self.g(lm + 'net.net.0.weight'), self.din, out_mode=1))
            self.gemm16(r, self.din, h, w['dH_cb'], wb['0T'], dptr(w['dX']), self.ldx)
        else:
            # layer 0 in fp32 (see _forward_nets_b16)
            dW(ev, lambda: self._linear_bwd_params(r, dptr(w['X']), self.ldx, self.din, dptr(w['dH_c']), h, h,
                                                   self.g(lm + 'net.net.0.weight'), None))
            self.gemm(0, 0, r, self.din, h, dptr(w['dH_c']), h, self.p(lm + 'net.net.0.weight'), self.din, dptr(w['dX']), self.ldx,
                      dense=True)
        if bucketed:
            self.flush_colsums()
        self.phase_bwd(w, N, view_idx, frame_idx, raw_phase, with_colsums=not bucketed)
        flush_dW()
        if aside:
            main.wait_stream(side)

    # ---- fp32-equivalent split-precision chain (self.mlp_split; csrc/gemm_xp.h) -------------------------------------------------
    def _use_xp(self, w):
        return self.mlp_split and 'Xx' in w

    def _use_xp_vp(self, w):
        return self.mlp_split and 'AAx' in w

    def _vposer_xp_weights(self):
        """xp copies of the frozen (BatchNorm-folded, fold_vposer) VPoser weights: plain [out][in] for the forward products, transposed
        [in][out] for the two products of the KL term's adjoint; fmt 2: their scale records (and the biases') -- all static."""
        if hasattr(self, '_vpx'):
            return
        vp = self.vp
        i16 = dict(dtype=torch.int16, device=self.device)
        xl = lambda k: int(self.lib.nemo_xp_ld(self.xp_fmt, k))
        self._vpmeta = torch.zeros(len(self.VPMETA), 64, dtype=torch.float32, device=self.device)
        vx, items = {}, []
        for name, T in (('e2w', True), ('d0mw', False), ('d3w', False), ('d5w', False), ('emw', True)):
            W = vp[name]
            rows, cols = W.shape
            vx[name] = torch.zeros(rows, xl(cols), **i16)
            if T:
                vx[name + 'T'] = torch.zeros(cols, xl(rows), **i16)
            items.append((dptr(W), rows, cols, W.stride(0), vx[name], vx.get(name + 'T'), self._vm(name)))
        if self.xp_fmt == 2:
            am = [(src, rows, cols, lds, meta) for src, rows, cols, lds, _, _, meta in items]
            am += [(dptr(vp[b]), 1, vp[b].numel(), vp[b].numel(), self._vm(b)) for b in ('e2b', 'd0mb', 'd3b')]
            self.absmax_xp(am, overwrite=True)
        self.cast_xp(items)
        self._vpx = vx

    def _vm(self, name):
        """device pointer of the (static) scale record of a VPoser weight / bias (fmt 2)"""
        if self.xp_fmt != 2:
            return None
        return self._vpmeta[self.VPMETA.index(name)].data_ptr()

    def gemm_xp(self, M, N, K, Ax, Bx, Cp=None, ldc=0, bias=None, act=0, maskx=None, alpha=1.0, out_mode=0, Cx=None, CxT=None,
                colsum=None, tag=None, mA=None, mB=None, mBias=None, mOut=None, mZero=None, mask_mode=1):
        """C (M x N, fp32, may be None) (op)= epilogue(alpha * A B^T), A (M x K) / B (N x K) xp matrices; Cx / CxT: the result's xp
        copies for the next products of the chain (include/nemo_hip.h nemo_gemm_xp).  mA / mB / mBias / mOut: the scale records of
        the operands, the bias and the result's copies (device pointers; fmt 2 only)."""
        ev = self._event_begin(tag, 2.0 * M * N * K)
        cur = torch.cuda.current_stream()
        ws = self.gemm_ws[1 if cur == self.side_stream else (2 if cur == self.side_stream2 else 0)]
        check(self.lib.nemo_gemm_xp(self.xp_fmt, M, N, K, Ax.data_ptr(), Ax.stride(0), Bx.data_ptr(), Bx.stride(0), Cp, ldc, bias, act,
                                    dptr(maskx), maskx.stride(0) if maskx is not None else 0, mask_mode if maskx is not None else 0, alpha,
                                    out_mode, dptr(Cx), Cx.stride(0) if Cx is not None else 0, dptr(CxT),
                                    CxT.stride(0) if CxT is not None else 0, 1.0, dptr(colsum),
                                    colsum.stride(0) if colsum is not None else 0, mA, mB, mBias, mOut, mZero, ws.data_ptr(),
                                    ws.numel() * 4, _stream()),
              'nemo_gemm_xp')
        self._event_end(ev)

    def absmax_xp(self, items, overwrite=False):
        """items: (src_ptr, rows, cols, lds, meta pointer) -- max |src| into their scale records, one launch (fmt 2); ``overwrite``:
        the records' slots are stored, not accumulated into (they need no zeroing)."""
        am = self._absmax_descs(items, overwrite)
        check(self.lib.nemo_absmax_multi(len(items), am, _stream()), 'nemo_absmax_multi')

    @staticmethod
    def _absmax_descs(items, overwrite):
        am = (_lib.AbsmaxDesc * len(items))()
        for i, (src, rows, cols, lds, meta) in enumerate(items):
            am[i].src, am[i].rows, am[i].cols, am[i].lds, am[i].meta, am[i].overwrite = src, rows, cols, lds, meta, int(overwrite)
        return am

    def cast_xp(self, items):
        """items: (src_ptr, rows, cols, lds, dst or None, dstT or None, meta pointer or None) -- their xp copies in ONE launch
        (fmt 2: scaled from the absmax their records hold)."""
        n = len(items)
        arr = (_lib.CastXpDesc * n)()
        for i, (src, rows, cols, lds, dst, dstT, meta) in enumerate(items):
            q = arr[i]
            q.src, q.rows, q.cols, q.lds, q.scale = src, rows, cols, lds, 1.0
            q.dst, q.ldd = dptr(dst), dst.stride(0) if dst is not None else 0
            q.dstT, q.lddT = dptr(dstT), dstT.stride(0) if dstT is not None else 0
            q.meta = meta if self.xp_fmt == 2 else None
        check(self.lib.nemo_cast_xp(self.xp_fmt, n, arr, _stream()), 'nemo_cast_xp')

    def _xm(self, w, name):
        """device pointer of the scale record of copy `name` (fmt 2; None for the bf16 form, which has no scales)"""
        if self.xp_fmt != 2:
            return None
        if name == 'X':
            return w['xmeta_x'].data_ptr()
        if name[0] in 'Wb':             # weights and biases: engine-level records, overwritten by every pass's absmax launch
            if not hasattr(self, '_wmeta'):
                self._wmeta = torch.zeros(len(self.XMETA), 64, dtype=torch.float32, device=self.device)
            return self._wmeta[self.XMETA.index(name)].data_ptr()
        return w['xmeta'][self.XMETA.index(name)].data_ptr()

    def head_meta(self, w, covers_all):
        """The scale record nemo_pose_bwd_fused should accumulate the head gradient's absmax into, or None.  ``covers_all``: the
        launch sees every column the backward will read (no translation columns, or it is handed them); otherwise the chain
        runs a pass of its own over dHEAD."""
        self._dhead_absmax_done = None
        if not (self.version >= 1 and self._use_xp(w) and self.xp_fmt == 2 and covers_all):
            return None
        self._dhead_absmax_done = w
        return self._xm(w, 'dHEAD')

    def _weight_items(self, w, transposed):
        """the cast descriptors of the four MotionNet weight matrices (plain [out][in] for the forward and -- `transposed` -- [in][out]
        for the activation gradients); they change in Adam at the step's end, so the copies are made every step"""
        lm, h = 'learned_motion.', self.h
        if not hasattr(self, '_wx'):
            i16 = dict(dtype=torch.int16, device=self.device)
            xl = lambda k: int(self.lib.nemo_xp_ld(self.xp_fmt, k))
            self._wx = {'0': torch.zeros(h, xl(self.din), **i16), '2': torch.zeros(h, xl(h), **i16), '4': torch.zeros(h, xl(h), **i16),
                        'head': torch.zeros(147, xl(h), **i16), '0T': torch.zeros(self.din, xl(h), **i16),
                        '2T': torch.zeros(h, xl(h), **i16), '4T': torch.zeros(h, xl(h), **i16), 'headT': torch.zeros(h, xl(147), **i16)}
        wx = self._wx
        T = lambda k: wx[k + 'T'] if transposed else None
        return [(self.p(lm + 'net.net.0.weight'), h, self.din, self.din, wx['0'], T('0'), self._xm(w, 'W0')),
                (self.p(lm + 'net.net.2.weight'), h, h, h, wx['2'], T('2'), self._xm(w, 'W2')),
                (self.p(lm + 'net.net.4.weight'), h, h, h, wx['4'], T('4'), self._xm(w, 'W4')),
                (self.p(lm + 'rot_out.weight'), 147, h, h, wx['head'], T('head'), self._xm(w, 'Whead'))]

    # ---- are the weights' absmax records (fmt 2) those of the CURRENT weights?  Host-side bookkeeping: every Adam launch through this
    # engine clears the flag, every pass over the weights sets it; writes from outside (load_state_dict, user code) are seen through the
    # flat buffer's version counter.  MultiViewModel._captured keeps the flag right across graph replays (a replay runs no Python).
    _wrec_ok, _wrec_version = False, -1

    def wrec_ok(self):
        return self._wrec_ok and self._wrec_version == self.params._version

    def refresh_weight_records(self, w):
        """The pass over the weights in front of a replayed graph that was captured while the records were fresh (an evaluation pass
        behind another one) and finds them stale."""
        if self.version >= 1 and self._use_xp(w) and self.xp_fmt == 2:
            self._weights_absmax(w, self._weight_items(w, False))
            self._wrec_ok, self._wrec_version = True, self.params._version

    def _weights_absmax(self, w, items):
        """fmt 2: the absmax records of the weights and of the three hidden-layer biases (which enter the bounds the hidden copies
        are scaled by) in one launch"""
        lm, h = 'learned_motion.', self.h
        self.absmax_xp(self._weights_absmax_items(w, items), overwrite=True)

    def _weights_absmax_items(self, w, items):
        lm, h = 'learned_motion.', self.h
        am = [(src, rows, cols, lds, meta) for src, rows, cols, lds, _, _, meta in items]
        am += [(self.p(lm + nm), 1, h, h, self._xm(w, key)) for nm, key in (('net.net.0.bias', 'b0'), ('net.net.2.bias', 'b2'),
                                                                                   ('net.net.4.bias', 'b4'))]
        return am

    def _forward_nets_xp(self, w, N, train, am_done=None):
        """forward_pose's MLP on the split-precision chain: every layer reads the previous layer's xp copy and leaves its own (plain
        for the next layer, transposed for its parameter gradient when `train`); the hidden activations exist as xp copies only,
        the heads' output in fp32.  Launches in front of the first product: [phase kernel (+ max |X|) on the main stream beside the
        weights' absmax pass on the side stream (fmt 2)] -> ONE cast launch (weights + X)."""
        lm, h, r = 'learned_motion.', self.h, N + 1
        main, side = torch.cuda.current_stream(), self.side_stream
        items = self._weight_items(w, bool(train))
        wx = self._wx
        T = lambda k: w[k] if train else None
        m = lambda k: self._xm(w, k)
        if am_done is not None:
            main.wait_event(am_done)        # (the weights' absmax pass forward_pose started on the side stream beside the phase kernel)
        self.cast_xp(items + [(dptr(w['X']), r, self.din, self.ldx, w['Xx'], T('XxT'), m('X'))])
        self._wxT_fresh = w if train else None
        self.gemm_xp(r, h, self.din, w['Xx'], wx['0'], bias=self.p(lm + 'net.net.0.bias'), act=1, Cx=w['H1x'], CxT=T('H1xT'),
                     mA=m('X'), mB=m('W0'), mBias=m('b0'), mOut=m('H1'))
        self.gemm_xp(r, h, h, w['H1x'], wx['2'], bias=self.p(lm + 'net.net.2.bias'), act=1, Cx=w['H2x'], CxT=T('H2xT'),
                     tag='gemm_mlp_hidden_fwd', mA=m('H1'), mB=m('W2'), mBias=m('b2'), mOut=m('H2'))
        self.gemm_xp(r, h, h, w['H2x'], wx['4'], bias=self.p(lm + 'net.net.4.bias'), act=1, Cx=w['H3x'], CxT=T('H3xT'),
                     mA=m('H2'), mB=m('W4'), mBias=m('b4'), mOut=m('H3'))
        # (the last product of the forward returns X's absmax slots to zero for the next pass's phase kernel)
        self.gemm_xp(r, 147, h, w['H3x'], wx['head'], dptr(w['HEAD']), HEAD_LD, bias=self.p(lm + 'rot_out.bias'), mA=m('H3'), mB=m('Whead'),
                     mZero=m('X'))

    def _dw_problem(self, grouped, M, N, K, Ax, Bx, Cp, ldc, mA, mB):
        """A parameter-gradient product of the chain: a problem record for the grouped launch, or the closure that launches it alone."""
        if grouped:
            return (M, N, K, Ax.data_ptr(), Ax.stride(0), Bx.data_ptr(), Bx.stride(0), Cp, ldc, 1.0, 1, mA, mB)
        return lambda: self.gemm_xp(M, N, K, Ax, Bx, Cp, ldc, out_mode=1, mA=mA, mB=mB)

    def _backward_mlp_xp(self, w, N, view_idx, frame_idx, raw_phase, nout, nbias, stages=(0, 1, 2), bucketed=False):
        """backward_mlp on the split-precision chain.  Every product is C = A B^T over xp copies: dX_l = dY_l (W_l^T)^T reads dY_l's
        plain copy and the transposed weight copy; dW_l = dY_l^T X_l reads the two TRANSPOSED activation copies (K = rows).
        ``stages`` / ``bucketed`` as in backward_mlp."""
        L, lm, h, r = self.lib, 'learned_motion.', self.h, N + 1
        if 0 in stages and getattr(self, '_wxT_fresh', None) is not w:
            items = self._weight_items(w, True)     # (no training forward of this workspace ran in front: warm-up after an eval pass)
            if self.xp_fmt == 2:
                self._weights_absmax(w, items)
            self.cast_xp(items)
        self._wxT_fresh = None
        wx = self._wx
        cs = self._colsums
        R = int(L.nemo_gemm_colsum_rows(r))
        m = lambda k: self._xm(w, k)

        def end_of_stage():
            if bucketed:
                self.flush_colsums()

        # the parameter-gradient products go beside the dX chain on the side stream (un-bucketed), each as soon as its dY exists,
        # ENQUEUED BEHIND the chain's next product (a replayed graph keeps a node's first successor on its queue)
        main, side = torch.cuda.current_stream(), self.side_stream
        aside = (not bucketed) and r <= self.XP_DW_ASIDE_ROWS
        pend = []
        # NEMO_XP_GROUPED_DW=1 (A/B aid, measured and NOT the default): the four parameter gradients as ONE grouped launch
        # (nemo_gemm_xp_grouped) on the side stream behind the dX chain's last hidden product instead of one launch each beside the dX
        # launches.  Same box, 8 x 300: 1.035 - 1.036 against 1.024 ms per step -- the dX launches alone are not faster in the graph (38 - 40
        # us, their operands come cold from the previous launch) and the grouped launch takes 78 us (profiles/r06_experiments.md).
        grouped = aside and tuple(stages) == (0, 1, 2) and r <= self.XP_GROUPED_DW_ROWS and os.environ.get('NEMO_XP_GROUPED_DW', '0') == '1'
        probs = []

        def ready():
            return main.record_event() if (aside and not grouped) else None

        def dW(ev, fn):
            if grouped:
                return probs.append(fn)
            if not aside:
                return fn()
            pend.append((ev, fn))

        def flush_dW():
            for ev, fn in pend:
                side.wait_event(ev)
                with torch.cuda.stream(side):
                    fn()
            del pend[:]

        if 0 in stages:
            if self.xp_fmt == 2 and getattr(self, '_dhead_absmax_done', None) is not w:
                self.absmax_xp([(dptr(w['dHEAD']), r, nout, HEAD_LD, m('dHEAD'))])       # (else: left by nemo_pose_bwd_fused)
            self._dhead_absmax_done = None
            self.cast_xp([(dptr(w['dHEAD']), r, nout, HEAD_LD, w['dHEADx'], w['dHEADxT'], m('dHEAD'))])
            # heads
            ev = ready()
            dW(ev, self._dw_problem(grouped, nout, h, r, w['dHEADxT'], w['H3xT'], self.g(lm + 'rot_out.weight'), h, m('dHEAD'), m('H3')))
            cs.append((dptr(w['dHEAD']), r, nbias, HEAD_LD, self.g(lm + 'rot_out.bias')))
            self.gemm_xp(r, h, nout, w['dHEADx'], wx['headT'], maskx=w['H3x'], Cx=w['dHx'], CxT=w['dHxT'], colsum=w['cs4'],
                         mA=m('dHEAD'), mB=m('Whead'), mOut=m('dH'))
            flush_dW()
            # layer 4
            ev = ready()
            dW(ev, self._dw_problem(grouped, h, h, r, w['dHxT'], w['H2xT'], self.g(lm + 'net.net.4.weight'), h, m('dH'), m('H2')))
            cs.append((dptr(w['cs4']), R, h, h, self.g(lm + 'net.net.4.bias')))
            self.gemm_xp(r, h, h, w['dHx'], wx['4T'], maskx=w['H2x'], Cx=w['dH_bx'], CxT=w['dH_bxT'], colsum=w['cs2'],
                         tag='gemm_mlp_hidden_dx', mA=m('dH'), mB=m('W4'), mOut=m('dH_b'))
            flush_dW()
            end_of_stage()
        if 1 in stages:
            # layer 2
            ev = ready()
            dW(ev, self._dw_problem(grouped, h, h, r, w['dH_bxT'], w['H1xT'], self.g(lm + 'net.net.2.weight'), h, m('dH_b'), m('H1')))
            cs.append((dptr(w['cs2']), R, h, h, self.g(lm + 'net.net.2.bias')))
            self.gemm_xp(r, h, h, w['dH_bx'], wx['2T'], maskx=w['H1x'], Cx=w['dH_cx'], CxT=w['dH_cxT'], colsum=w['cs0'],
                         mA=m('dH_b'), mB=m('W2'), mOut=m('dH_c'))
            flush_dW()
            end_of_stage()
        if 2 not in stages:
            return
        ev = ready()
        cs.append((dptr(w['cs0']), R, h, h, self.g(lm + 'net.net.0.bias')))
        dW(ev, self._dw_problem(grouped, h, self.din, r, w['dH_cxT'], w['XxT'], self.g(lm + 'net.net.0.weight'), self.din, m('dH_c'), m('X')))
        if grouped:
            # every dY exists: the four gradients in one launch on the side stream, enqueued BEHIND the chain's next product
            ev_g = main.record_event()
        self.gemm_xp(r, self.din, h, w['dH_cx'], wx['0T'], dptr(w['dX']), self.ldx, mA=m('dH_c'), mB=m('W0'))
        if grouped:
            side.wait_event(ev_g)
            with torch.cuda.stream(side):
                arr = (_lib.GemmXpProblem * len(probs))()
                for i, q in enumerate(probs):
                    (arr[i].M, arr[i].N, arr[i].K, arr[i].A, arr[i].lda, arr[i].B, arr[i].ldb, arr[i].C, arr[i].ldc, arr[i].alpha,
                     arr[i].out_mode, arr[i].metaA, arr[i].metaB) = q
                gws = self.gemm_ws[1]
                check(self.lib.nemo_gemm_xp_grouped(self.xp_fmt, len(probs), arr, gws.data_ptr(), gws.numel() * 4, _stream()),
                      'nemo_gemm_xp_grouped')
        if bucketed:
            self.flush_colsums()
        self.phase_bwd(w, N, view_idx, frame_idx, raw_phase, with_colsums=not bucketed)
        flush_dW()
        if aside:
            main.wait_stream(side)

    def gemm_grouped(self, problems, dense=True):
        """problems: list of (ta, tb, M, N, K, A, lda, B, ldb, C, ldc, out_mode) -- independent products of one layout in
        ONE launch (nemo_gemm_grouped_f32: the parameter gradients of the whole MLP backward)."""
        arr = (_lib.GemmProblem * len(problems))()
        for i, (ta, tb, M, N, K, A, lda, B, ldb, Cp, ldc, om) in enumerate(problems):
            q = arr[i]
            q.transA, q.transB, q.M, q.N, q.K = ta, tb, M, N, K
            q.A, q.lda, q.B, q.ldb, q.C, q.ldc, q.alpha, q.out_mode = A, lda, B, ldb, Cp, ldc, 1.0, om
        cur = torch.cuda.current_stream()
        ws = self.gemm_ws[1 if cur == self.side_stream else (2 if cur == self.side_stream2 else 0)]
        fn = self.lib.nemo_gemm_grouped_bf16 if (dense and self.bf16) else self.lib.nemo_gemm_grouped_f32
        check(fn(len(problems), arr, ws.data_ptr(), ws.numel() * 4, _stream()), 'nemo_gemm_grouped')

    def phase_bwd(self, w, N, view_idx, frame_idx, raw_phase, with_colsums=False):
        """Phase / RBF / code backward; ``with_colsums``: the pending bias column sums (self._colsums) ride in further blocks
        of the same launch (nemo_phase_embed_bwd_colsum) instead of a launch of their own."""
        args = (N, self.V, self.T, self.K, self.D, self.C, dptr(view_idx), dptr(frame_idx), dptr(raw_phase),
                self.p('phase_networks.0.shifts'), self.p('phase_networks.0.scales'), self.ldp,
                self.p('phase_rbf.log_sigmas') if self.D > 0 else None, self.kernel_id, dptr(w['phase']),
                dptr(w['dX']), self.ldx, dptr(w['phase_ws']), self.g('phase_networks.0.shifts'),
                self.g('phase_networks.0.scales'),
                self.g('phase_rbf.log_sigmas') if self.D > 0 else None,
                self.g('learned_instance_code') if self.C > 0 else None)
        if with_colsums and self._colsums:
            n = len(self._colsums)
            arr = (ColsumDesc * n)()
            for i, (x, m, nn, ld, out) in enumerate(self._colsums):
                arr[i].X, arr[i].M, arr[i].N, arr[i].ldx, arr[i].out = x, m, nn, ld, out
            self._colsums = []
            check(self.lib.nemo_phase_embed_bwd_colsum(*args, n, arr, int(self.batch_sorted), _stream()), 'nemo_phase_embed_bwd_colsum')
        else:
            check(self.lib.nemo_phase_embed_bwd(*args, int(self.batch_sorted), _stream()), 'nemo_phase_embed_bwd')

    def flush_colsums(self):
        if not self._colsums:
            return
        n = len(self._colsums)
        arr = (ColsumDesc * n)()
        for i, (x, m, nn, ld, out) in enumerate(self._colsums):
            arr[i].X, arr[i].M, arr[i].N, arr[i].ldx, arr[i].out = x, m, nn, ld, out
        check(self.lib.nemo_colsum_multi(n, arr, _stream()), 'nemo_colsum_multi')
        self._colsums = []

    # ------------------------------------------------------------------ forward pieces
    def forward_pose(self, w, N, view_idx, frame_idx, raw_phase=None, code_noise=None, train=True, begin=None):
        """K1-K5: phase warp, RBF, MLP, rot6d->R->aa.  Fills X,H1..H3,ROT,TR,R,AA.
        ``begin`` = (arena, zero_grads, n_seg): what ``step_begin`` would be called with -- done by further blocks of the
        phase kernel's launch instead of a launch of its own (nemo_phase_embed_fwd_begin)."""
        # every pass starts here: the launches that follow take their regions of this engine's ordered-reduction arena from its start
        # (include/nemo_hip.h nemo_reduce_ws_bind: deterministic sums instead of float atomics)
        self.bind_reduce_ws()
        L, st = self.lib, _stream()
        sh0 = self.p('phase_networks.0.shifts')
        sc0 = self.p('phase_networks.0.scales')
        pargs = (N, self.V, self.T, self.K, self.D, self.C, dptr(view_idx), dptr(frame_idx), dptr(raw_phase),
                 sh0, sc0, self.ldp, self.p('phase_rbf.log_sigmas') if self.D > 0 else None,
                 self.p('learned_instance_code') if self.C > 0 else None, dptr(code_noise), self.kernel_id,
                 dptr(w['X']), self.ldx, dptr(w['phase']), dptr(w['phase_ws']),
                 # (fmt 2 of the split-precision chain: the phase kernel leaves max |X| in X's scale record)
                 w['xmeta_x'].data_ptr() if (self.version >= 1 and self._use_xp(w) and self.xp_fmt == 2) else None)
        am_done, am_n, am_descs = None, 0, None
        if self.version >= 1 and self._use_xp(w) and self.xp_fmt == 2 and not self.wrec_ok():
            # the weights' / biases' absmax records do not describe the current weights (first pass, an update whose launch did not
            # refresh them, a checkpoint load): the pass over the weights runs in the leading blocks of the step's first launch
            # (nemo_phase_embed_fwd_begin, ABI 18) -- or, for a pass without that launch (evaluation) or with NEMO_ABSMAX_ASIDE=1 (the
            # form before: A/B aid), on the side stream beside the phase kernel (14 us there, the phase kernel 25 instead of 10, and
            # a fork + cross-queue join in front of the chain's cast launch).
            am_items = self._weights_absmax_items(w, self._weight_items(w, bool(train)))
            if begin is not None and os.environ.get('NEMO_ABSMAX_ASIDE') != '1':
                am_n, am_descs = len(am_items), self._absmax_descs(am_items, True)
            else:
                main, side = torch.cuda.current_stream(), self.side_stream
                side.wait_event(main.record_event())
                with torch.cuda.stream(side):
                    self.absmax_xp(am_items, overwrite=True)
                    am_done = side.record_event()
            self._wrec_ok, self._wrec_version = True, self.params._version
        if begin is not None:
            arena, zero_grads, n_seg = begin
            check(L.nemo_phase_embed_fwd_begin(*pargs, arena.data_ptr(), arena.numel() * 4,
                                               self.grads.data_ptr() if zero_grads else None,
                                               self.grads.numel() * 4 if zero_grads else 0,
                                               self._seg_dev.data_ptr() if n_seg else None, n_seg, 0.9, 0.999, am_n, am_descs, st),
                  'nemo_phase_embed_fwd_begin')
        else:
            check(L.nemo_phase_embed_fwd(*pargs, st), 'nemo_phase_embed_fwd')
        h, r = self.h, N + 1
        if self.version == 0:
            return self._forward_nets_v0(w, N)
        lm = 'learned_motion.'
        if self.b16mem:
            self._forward_nets_b16(w, N, train)
            check(L.nemo_rot6d_fwd(N, 24, dptr(w['ROT']), HEAD_LD, 1, dptr(w['R']), dptr(w['AA']), st), 'nemo_rot6d_fwd')
            return
        if self._use_xp(w):
            self._forward_nets_xp(w, N, train, am_done)
            check(L.nemo_rot6d_fwd(N, 24, dptr(w['ROT']), HEAD_LD, 1, dptr(w['R']), dptr(w['AA']), st), 'nemo_rot6d_fwd')
            return
        self._linear(r, dptr(w['X']), self.ldx, self.din, self.p(lm + 'net.net.0.weight'),
                     self.p(lm + 'net.net.0.bias'), h, dptr(w['H1']), h, act=1)
        self._linear(r, dptr(w['H1']), h, h, self.p(lm + 'net.net.2.weight'), self.p(lm + 'net.net.2.bias'),
                     h, dptr(w['H2']), h, act=1, tag='gemm_mlp_hidden_fwd')
        self._linear(r, dptr(w['H2']), h, h, self.p(lm + 'net.net.4.weight'), self.p(lm + 'net.net.4.bias'),
                     h, dptr(w['H3']), h, act=1)
        # both heads in one GEMM: [rot_out.weight ; linear_out.weight] are adjacent in the flat buffer
        self._linear(r, dptr(w['H3']), h, h, self.p(lm + 'rot_out.weight'), self.p(lm + 'rot_out.bias'),
                     147, dptr(w['HEAD']), HEAD_LD)
        check(L.nemo_rot6d_fwd(N, 24, dptr(w['ROT']), HEAD_LD, 1, dptr(w['R']), dptr(w['AA']), st),
              'nemo_rot6d_fwd')

    # ---- NemoV0: three networks on the warped phase (:3005-3034).  Their outputs land in the columns of the merged head
    # buffer the later NemoV* write ([orient 6 | poses 138 | trans 3]), so everything downstream is shared.
    V0_NETS = (('learned_poses', 6, 138, ('H1', 'H2', 'H3')), ('learned_orient', 0, 6, ('O1', 'O2', 'O3')))

    def _forward_nets_v0(self, w, N):
        h, r, X = self.h, N + 1, dptr(w['X'])
        head = w['HEAD'].data_ptr()
        for net, col, nout, (a1, a2, a3) in self.V0_NETS:
            q = net + '.net.net.'
            self._linear(r, X, self.ldx, 1, self.p(q + '0.weight'), self.p(q + '0.bias'), h, dptr(w[a1]), h, act=1)
            self._linear(r, dptr(w[a1]), h, h, self.p(q + '2.weight'), self.p(q + '2.bias'), h, dptr(w[a2]), h, act=1)
            self._linear(r, dptr(w[a2]), h, h, self.p(q + '4.weight'), self.p(q + '4.bias'), h, dptr(w[a3]), h, act=1)
            self._linear(r, dptr(w[a3]), h, h, self.p(net + '.linear.weight'), self.p(net + '.linear.bias'), nout,
                         head + 4 * col, HEAD_LD)
        q = 'learned_trans.net.'
        self._linear(r, X, self.ldx, 1, self.p(q + '0.weight'), self.p(q + '0.bias'), h, dptr(w['T1']), h, act=1)
        self._linear(r, dptr(w['T1']), h, h, self.p(q + '2.weight'), self.p(q + '2.bias'), h, dptr(w['T2']), h, act=1)
        self._linear(r, dptr(w['T2']), h, h, self.p(q + '4.weight'), self.p(q + '4.bias'), 3, head + 4 * 144, HEAD_LD)
        check(self.lib.nemo_rot6d_fwd(N, 24, dptr(w['ROT']), HEAD_LD, 1, dptr(w['R']), dptr(w['AA']), _stream()),
              'nemo_rot6d_fwd')

    def _backward_nets_v0(self, w, N, view_idx, frame_idx, raw_phase, has_trans_grad=True):
        """dHEAD (N+1, [orient 6 | poses 138 | trans 3]) -> gradients of the three networks and of the phase networks."""
        h, r, X = self.h, N + 1, dptr(w['X'])
        dhead = w['dHEAD'].data_ptr()
        first = True
        for net, col, nout, (a1, a2, a3) in self.V0_NETS:
            q = net + '.net.net.'
            dY = dhead + 4 * col
            self._linear_bwd_params(r, dptr(w[a3]), h, h, dY, HEAD_LD, nout, self.g(net + '.linear.weight'),
                                    self.g(net + '.linear.bias'))
            self.gemm(0, 0, r, h, nout, dY, HEAD_LD, self.p(net + '.linear.weight'), h, dptr(w['dH']), h,
                      mask=dptr(w[a3]), ldmask=h, mask_mode=1, dense=True)
            self._linear_bwd_params(r, dptr(w[a2]), h, h, dptr(w['dH']), h, h, self.g(q + '4.weight'), self.g(q + '4.bias'))
            self.gemm(0, 0, r, h, h, dptr(w['dH']), h, self.p(q + '4.weight'), h, dptr(w['dH_b']), h,
                      mask=dptr(w[a2]), ldmask=h, mask_mode=1, dense=True)
            self._linear_bwd_params(r, dptr(w[a1]), h, h, dptr(w['dH_b']), h, h, self.g(q + '2.weight'), self.g(q + '2.bias'))
            self.gemm(0, 0, r, h, h, dptr(w['dH_b']), h, self.p(q + '2.weight'), h, dptr(w['dH_c']), h,
                      mask=dptr(w[a1]), ldmask=h, mask_mode=1, dense=True)
            self._linear_bwd_params(r, X, self.ldx, 1, dptr(w['dH_c']), h, h, self.g(q + '0.weight'), self.g(q + '0.bias'))
            self.gemm(0, 0, r, 1, h, dptr(w['dH_c']), h, self.p(q + '0.weight'), 1, dptr(w['dX']), self.ldx,
                      out_mode=0 if first else 1)
            first = False
            self.flush_colsums()            # (dH .. dH_c are reused by the next network)
        if has_trans_grad:
            q = 'learned_trans.net.'
            dY = dhead + 4 * 144
            # trans - trans_0 cancels the last bias exactly (the reference's autograd yields an exact 0 there)
            self._linear_bwd_params(r, dptr(w['T2']), h, h, dY, HEAD_LD, 3, self.g(q + '4.weight'),
                                    self.g(q + '4.bias') if self.start_global_traj_anywhere else None)
            self.gemm(0, 0, r, h, 3, dY, HEAD_LD, self.p(q + '4.weight'), h, dptr(w['dH']), h,
                      mask=dptr(w['T2']), ldmask=h, mask_mode=1)
            self._linear_bwd_params(r, dptr(w['T1']), h, h, dptr(w['dH']), h, h, self.g(q + '2.weight'), self.g(q + '2.bias'))
            self.gemm(0, 0, r, h, h, dptr(w['dH']), h, self.p(q + '2.weight'), h, dptr(w['dH_b']), h,
                      mask=dptr(w['T1']), ldmask=h, mask_mode=1, dense=True)
            self._linear_bwd_params(r, X, self.ldx, 1, dptr(w['dH_b']), h, h, self.g(q + '0.weight'), self.g(q + '0.bias'))
            self.gemm(0, 0, r, 1, h, dptr(w['dH_b']), h, self.p(q + '0.weight'), 1, dptr(w['dX']), self.ldx, out_mode=1)
            self.flush_colsums()
        check(self.lib.nemo_phase_embed_bwd(
            N, self.V, self.T, self.K, 0, 0, dptr(view_idx), dptr(frame_idx), dptr(raw_phase),
            self.p('phase_networks.0.shifts'), self.p('phase_networks.0.scales'), self.ldp, None, 0,
            dptr(w['phase']), dptr(w['dX']), self.ldx, dptr(w['phase_ws']), self.g('phase_networks.0.shifts'),
            self.g('phase_networks.0.scales'), None, None, int(self.batch_sorted), _stream()), 'nemo_phase_embed_bwd')

    def forward_joints(self, w, N, view_idx, frame_idx, with_loss, mean_mode=0, add_trans=True, ctx=None,
                       j3d=None, p2d=None, finalize=True):
        """K6-K8: FK, mesh-functional joints, projection, 2-D loss accumulators.  ``finalize=False``: the caller runs
        ``finalize_kp`` itself (off the main chain: ``backward_kp(norm_from_acc=True)`` does not need its output)."""
        Mq = self.joint_functionals(w, N, ctx)
        self.project_and_loss(w, N, view_idx, frame_idx, Mq, mean_mode, with_loss, add_trans, ctx, j3d, p2d, finalize)
        return Mq

    def joint_functionals(self, w, N, ctx=None):
        """K6: FK of the N bodies and the pre-contracted mesh functionals Mq = PF C1 + c0 (nothing here depends on the
        cameras: a camera fit evaluates it once)."""
        L, st = self.lib, _stream()
        ctx = ctx or self.ctx
        self.sync_betas(ctx)
        check(L.nemo_fk_fwd(ctx.handle, N, dptr(w['R']), dptr(w['A']), dptr(w['Jp']), dptr(w['PF']), 208, st),
              'nemo_fk_fwd')
        nq72 = ctx.nq * 72
        Mq = w['Mq'] if ctx is self.ctx else torch.zeros(N, max(nq72, 1), device=self.device)
        if ctx.nq:
            self.gemm(0, 0, N, nq72, 207, dptr(w['PF']), 208, ctx.C1, nq72, dptr(Mq), max(nq72, 1),
                      bias=ctx.c0)
        return Mq

    def project_and_loss(self, w, N, view_idx, frame_idx, Mq, mean_mode=0, with_loss=True, add_trans=True, ctx=None,
                         j3d=None, p2d=None, finalize=True):
        """K7 + K8: output joints, camera projection, 2-D loss accumulators (+ the loss scalar when ``finalize``)."""
        L, st = self.lib, _stream()
        ctx = ctx or self.ctx
        nq72 = ctx.nq * 72
        lt = LOSS_TYPES[self.args.loss]
        check(L.nemo_kp_fwd(
            ctx.handle, N, self.V, self.T, dptr(w['A']), dptr(w['Jp']), dptr(Mq), max(nq72, 1),
            dptr(w['TR']), HEAD_LD, 1 if (add_trans and not self.start_global_traj_anywhere) else 0,
            dptr(view_idx), dptr(frame_idx), self.p('learned_cameras'),
            dptr(self.targets) if with_loss else None, dptr(self.gt_size) if with_loss else None,
            FOCAL_LENGTH, self.cx, self.cy, lt, mean_mode,
            dptr(j3d if j3d is not None else w['j3d']), dptr(p2d if p2d is not None else w['p2d']),
            dptr(w['loss_all']) if with_loss else None, dptr(w['view_acc']) if with_loss else None, self.nvalid, st),
            'nemo_kp_fwd')
        if with_loss and finalize:
            self.finalize_kp(w, mean_mode, ctx)

    def finalize_kp(self, w, mean_mode=0, ctx=None):
        """Per-view accumulators -> the keypoint loss scalar (+ the normaliser, for callers of nemo_kp_bwd that pass it)."""
        ctx = ctx or self.ctx
        Wd = 1 if LOSS_TYPES[self.args.loss] in (2, 3, 5) else 2
        check(self.lib.nemo_kp_finalize(self.V, ctx.n_out, Wd, mean_mode, dptr(w['view_acc']),
                                        self.scal.data_ptr() + 4 * S_KP, dptr(w['norm']), _stream()), 'nemo_kp_finalize')

    def sync_betas(self, ctx=None):
        """Host-side: push ``learned_betas`` into the SMPL context when the tensor has been written since the
        last push (checkpoint load, user assignment).  Costs a version compare in the steady state.  ``step()``
        calls it BEFORE replaying a captured graph -- a replay skips the Python body that used to do it."""
        ctx = ctx or self.ctx
        if self.betas._version != getattr(ctx, '_betas_version', None):   # no D2H sync in the steady state
            was = ctx.split_ok
            ctx.set_betas(self.betas.detach().cpu().numpy())
            ctx._betas_version = self.betas._version
            if ctx.split_ok != was:        # the mesh kernel variant changes: captured graphs bake the old one in
                for w_ in self.ws.values():
                    w_['graphs'].clear()

    def vposer_mulv(self, w, N):
        """The encoder's (mu | logvar) product: only the KL term reads it, so it runs on the KL stream (the decoder starts
        from the encoder's hidden activation, forward_vposer)."""
        vp = self.vp
        if self._use_xp_vp(w):
            self.gemm_xp(N, 64, 512, w['E1x'], self._vpx['emw'], dptr(w['MULV']), 64, bias=dptr(vp['emb']), mA=self._xm(w, 'vE1'),
                         mB=self._vm('emw'))
        elif self.b16mem:
            self.gemm16(N, 64, 512, w['E1b'], self.vpb['emw'], dptr(w['MULV']), 64, bias=dptr(vp['emb']), Cb=w['MULVb'])
        else:
            self._linear(N, dptr(w['E1']), 512, 512, dptr(vp['emw']), dptr(vp['emb']), 64, dptr(w['MULV']), 64)

    def forward_vposer(self, w, N):
        """K9: encode -> decode(mean) (vposer_model.py:100-113, :3569).  The decoder's first layer is composed with the mean head
        (fold_vposer: d0mw), so the decoder chain starts from the encoder's hidden activation E1 and the (mu | logvar)
        product leaves the chain the mesh kernel waits for (vposer_mulv, on the KL stream).  Returns the event recorded once
        E1 exists.  The decoder's 6-D output D3 becomes axis-angle inside forward_v2v_pre's launch for the first mesh chunk
        and here (nemo_rot6d_fwd) for the samples behind it."""
        st, vp = _stream(), self.vp
        aa63 = w['AA'].data_ptr() + 4 * 3
        cur = torch.cuda.current_stream()
        if self._use_xp_vp(w):
            # split precision (nemo_gemm_xp; >= VP_XP_MIN_ROWS rows): every layer reads the previous layer's xp copy and leaves its own;
            # the hidden activations exist as xp copies only (E1x is also the LeakyReLU' mask of the KL term's adjoint)
            vx, m, vm = self._vpx, (lambda k: self._xm(w, k)), self._vm
            if self.xp_fmt == 2:
                self.absmax_xp([(aa63, N, 63, 72, m('vAA'))], overwrite=True)
            self.cast_xp([(aa63, N, 63, 72, w['AAx'], None, m('vAA'))])
            self.gemm_xp(N, 512, 63, w['AAx'], vx['e2w'], bias=dptr(vp['e2b']), act=2, Cx=w['E1x'], mA=m('vAA'), mB=vm('e2w'),
                         mBias=vm('e2b'), mOut=m('vE1'))
            enc_done = cur.record_event()
            self.gemm_xp(N, 512, 512, w['E1x'], vx['d0mw'], bias=dptr(vp['d0mb']), act=2, Cx=w['D1x'], mA=m('vE1'), mB=vm('d0mw'),
                         mBias=vm('d0mb'), mOut=m('vD1'))
            self.gemm_xp(N, 512, 512, w['D1x'], vx['d3w'], bias=dptr(vp['d3b']), act=2, Cx=w['D2x'], mA=m('vD1'), mB=vm('d3w'),
                         mBias=vm('d3b'), mOut=m('vD2'))
            self.gemm_xp(N, 126, 512, w['D2x'], vx['d5w'], dptr(w['D3']), 126, bias=dptr(vp['d5b']), mA=m('vD2'), mB=vm('d5w'))
        elif self.b16mem:
            vb = self.vpb
            self._cast(N, 63, aa63, 72, w['AAb'], 0)
            self.gemm16(N, 512, 63, w['AAb'], vb['e2w'], dptr(w['E1']), 512, bias=dptr(vp['e2b']), act=2, Cb=w['E1b'])
            enc_done = cur.record_event()
            self.gemm16(N, 512, 512, w['E1b'], vb['d0mw'], dptr(w['D1']), 512, bias=dptr(vp['d0mb']), act=2, Cb=w['D1b'])
            self.gemm16(N, 512, 512, w['D1b'], vb['d3w'], dptr(w['D2']), 512, bias=dptr(vp['d3b']), act=2, Cb=w['D2b'])
            self.gemm16(N, 126, 512, w['D2b'], vb['d5w'], dptr(w['D3']), 126, bias=dptr(vp['d5b']))
        else:
            self._linear(N, aa63, 72, 63, dptr(vp['e2w_p']), dptr(vp['e2b']), 512, dptr(w['E1']), 512, act=2, ldw=64)
            enc_done = cur.record_event()
            self._linear(N, dptr(w['E1']), 512, 512, dptr(vp['d0mw']), dptr(vp['d0mb']), 512, dptr(w['D1']), 512, act=2)
            self._linear(N, dptr(w['D1']), 512, 512, dptr(vp['d3w']), dptr(vp['d3b']), 512, dptr(w['D2']), 512, act=2)
            self._linear(N, dptr(w['D2']), 512, 512, dptr(vp['d5w']), dptr(vp['d5b']), 126, dptr(w['D3']), 126)
        first = min(w['Nc'], N)
        if first < N:
            check(self.lib.nemo_rot6d_fwd(N - first, 21, w['D3'].data_ptr() + 4 * first * 126, 126, 0, None,
                                          w['AAdec'].data_ptr() + 4 * first * 63, st), 'nemo_rot6d_fwd')
        return enc_done

    def vposer_kl(self, w, N):
        """K12: KL( N(mu, softplus(logvar)) || N(0, 1) ) and its gradient w.r.t. (mu | logvar)."""
        check(self.lib.nemo_kl_fwd_bwd(N, 32, dptr(w['MULV']), 64, self.scal.data_ptr() + 4 * S_KL,
                                       dptr(w['dMULV']), 64, self.nvalid, _stream()), 'nemo_kl_fwd_bwd')

    def forward_v2v_pre(self, w, N):
        """The part of forward_v2v's first chunk that only needs the poses: both bodies' rotations -- incl. the decoder output's
        6-D -> axis-angle conversion (nemo_v2v_prep_fwd_dec) -- and their FK.  The step runs it at the end of the VPoser
        stream, off the main chain (``forward_v2v(pre_done=True)``)."""
        L, st, ctx = self.lib, _stream(), self.ctx
        n = min(w['Nc'], N)
        check(L.nemo_v2v_prep_fwd_dec(n, dptr(w['R']), dptr(w['AA']), dptr(w['D3']), 126, dptr(w['AAdec']), dptr(w['R2']),
                                      self.nvalid, st), 'nemo_v2v_prep_fwd_dec')
        check(L.nemo_fk_fwd(ctx.handle, 2 * n, dptr(w['R2']), dptr(w['A2']), dptr(w['Jp2']), dptr(w['PF2']), 208, st),
              'nemo_fk_fwd')

    def forward_v2v(self, w, N, need_grad, after_loss=None, pre_done=False):
        """K10 + K11: the two full-mesh bodies, L1 sum and (same pass) its gradient wrt the pose.
        One fused MFMA kernel per chunk (pose blend + skinning + L1 + gradient, nothing of the
        blended mesh goes to HBM); the blend-shape adjoint is one GEMM on the transposed dVP."""
        L, st, ctx = self.lib, _stream(), self.ctx
        Nc, NV3, ldP = w['Nc'], 3 * self.NV, self.ctx.ldP
        ldn = w['dVPt'].shape[1] if w['dVPt'].dim() == 2 else 0          # (bf16-in-memory chain: dVPb instead)
        for c0 in range(0, N, Nc):
            n = min(Nc, N - c0)
            R = w['R'].data_ptr() + 4 * c0 * 216
            AA = w['AA'].data_ptr() + 4 * c0 * 72
            AAd = w['AAdec'].data_ptr() + 4 * c0 * 63
            if not (pre_done and c0 == 0):
                check(L.nemo_v2v_prep_fwd(n, R, AA, AAd, dptr(w['R2']), self.nvalid if c0 == 0 else None, st), 'nemo_v2v_prep_fwd')
                check(L.nemo_fk_fwd(ctx.handle, 2 * n, dptr(w['R2']), dptr(w['A2']), dptr(w['Jp2']),
                                    dptr(w['PF2']), 208, st), 'nemo_fk_fwd')
            ev = self._event_begin('mesh_v2v_fused', 2.0 * n * self.NV * self.mesh_macs())
            ws = w['mesh_ws']
            fused = L.nemo_v2v_fused_bf16 if self.bf16 else (L.nemo_v2v_fused_split if self.mesh_split else L.nemo_v2v_fused)
            # single-chunk batches: the per-group sum of the blocks' partial dA images runs as a launch of its own on the
            # second side stream, beside the blend-shape adjoint GEMM (only the FK adjoint behind that GEMM needs dA)
            # (large batches only: at a one-instance shard the extra fork / join of the replayed graph costs more than the
            #  reduction's tail -- 0.509 against 0.503 ms, same box; 8 x 300: 1.557 against 1.564 ms; round 4: minibatch-512 steps
            #  0.572 against 0.579 ms deferred, two instances 0.602 against 0.618)
            defer = need_grad and self.SMALL_BATCH_ROWS < N <= Nc and self.timers is None
            # (round 6, the adjoint through nemo_gemm_xp takes whole CUs and the combine beside it 83 us instead of 25 -- the FK adjoint
            #  behind both starts ~16 us after the GEMM ends; still ahead of the in-kernel combine: 0.992 against 1.020 ms per step)
            # (fp16 piece planes of d vp only while the body model is inside the fp16 form's range: nemo_ctx_split_ok)
            hsplit = need_grad and self.split_adj and ctx.split_ok and (w['dVPh'] is not None or w['dVPx'] is not None) and n >= self.SPLIT_ADJ_ROWS
            hxp = hsplit and w['dVPx'] is not None
            if hxp:
                dx = w['dVPx']
                check(L.nemo_v2v_fused_splitxp(ctx.handle, n, dptr(w['PF2']), 208, dptr(w['A2']), self.scal.data_ptr() + 4 * S_V2V,
                                               dx.data_ptr(), dx.stride(0), None if defer else dptr(w['dA2']),
                                               ws.data_ptr(), ws.numel() * 4, st), 'nemo_v2v_fused_splitxp')
            elif hsplit:
                dh = w['dVPh']
                check(L.nemo_v2v_fused_splitmem(ctx.handle, n, dptr(w['PF2']), 208, dptr(w['A2']), self.scal.data_ptr() + 4 * S_V2V,
                                                dh.data_ptr(), dh.stride(1), dh.stride(0), None if defer else dptr(w['dA2']),
                                                ws.data_ptr(), ws.numel() * 4, st), 'nemo_v2v_fused_splitmem')
            elif self.b16mem:
                check(L.nemo_v2v_fused_bf16mem(ctx.handle, n, dptr(w['PF2']), 208, dptr(w['A2']),
                                               self.scal.data_ptr() + 4 * S_V2V, dptr(w['dVPb']), w['dVPb'].stride(0),
                                               None if defer else dptr(w['dA2']), ws.data_ptr(), ws.numel() * 4, st),
                      'nemo_v2v_fused_bf16mem')
            else:
                check(fused(ctx.handle, n, dptr(w['PF2']), 208, dptr(w['A2']),
                            self.scal.data_ptr() + 4 * S_V2V, dptr(w['dVPt']), ldn, None if defer else dptr(w['dA2']),
                            ws.data_ptr(), ws.numel() * 4, st), 'nemo_v2v_fused')
            self._event_end(ev)
            mesh_done = torch.cuda.current_stream().record_event() if defer else None
            if after_loss is not None and c0 + Nc >= N:
                after_loss()        # the L1 sum is final once the last chunk's mesh kernel has run
            if need_grad:
                if c0 > 0:
                    w['dPF2'].zero_()
                if hxp:
                    ev2 = self._event_begin('gemm_pose_blend_bwd', 2.0 * n * 207 * NV3)
                    cur = torch.cuda.current_stream()
                    gws = self.gemm_ws[1 if cur == self.side_stream else (2 if cur == self.side_stream2 else 0)]
                    dx = w['dVPx']
                    check(L.nemo_gemm_xp(2, n, 207, NV3, dx.data_ptr(), dx.stride(0), self.Px.data_ptr(), self.Px.stride(0), dptr(w['dPF2']), 208,
                                         None, 0, None, 0, 0, 1.0, 1, None, 0, None, 0, 1.0, None, 0, self.adj_meta[0].data_ptr(),
                                         self.adj_meta[1].data_ptr(), None, None, None, gws.data_ptr(), gws.numel() * 4, st), 'nemo_gemm_xp')
                    self._event_end(ev2)
                elif hsplit:
                    ev2 = self._event_begin('gemm_pose_blend_bwd', 2.0 * n * 207 * NV3)
                    cur = torch.cuda.current_stream()
                    gws = self.gemm_ws[1 if cur == self.side_stream else (2 if cur == self.side_stream2 else 0)]
                    dh = w['dVPh']
                    check(L.nemo_gemm_f16x2mem_adj(n, 207, (NV3 + 1) // 2 * 2, dh.data_ptr(), dh.stride(1), dh.stride(0),
                                                   self.Ph.data_ptr(), self.Ph.stride(1), self.Ph.stride(0), dptr(w['dPF2']), 208,
                                                   1.0 / (4096.0 * self.ph_scale), 1, gws.data_ptr(), gws.numel() * 4, st),
                          'nemo_gemm_f16x2mem_adj')
                    self._event_end(ev2)
                elif self.b16mem:
                    self.gemm16(n, 207, NV3, w['dVPb'], self.Pb, dptr(w['dPF2']), 208, out_mode=1, tag='gemm_pose_blend_bwd')
                else:
                    self.gemm(1, 1, n, 207, NV3, dptr(w['dVPt']), ldn, ctx.posedirs, ldP, dptr(w['dPF2']), 208,
                              out_mode=1, tag='gemm_pose_blend_bwd', dense=True)
                if defer:           # (enqueued AFTER the GEMM: a replayed graph keeps the first successor on the queue)
                    side2 = self.side_stream2
                    side2.wait_event(mesh_done)
                    with torch.cuda.stream(side2):
                        check(L.nemo_v2v_combine(ctx.handle, n, dptr(w['dA2']), ws.data_ptr(), ws.numel() * 4, _stream()),
                              'nemo_v2v_combine')
                    torch.cuda.current_stream().wait_stream(side2)
                check(L.nemo_fk_bwd(ctx.handle, n, dptr(w['R2']), dptr(w['A2']), dptr(w['dA2']), None,
                                    dptr(w['dPF2']), 208, w['dR2'].data_ptr() + 4 * c0 * 216, st), 'nemo_fk_bwd')

    # ------------------------------------------------------------------ backward pieces
    def backward_kp(self, w, N, view_idx, frame_idx, Mq, mean_mode, upstream, cams_only=False,
                    detach_pose=False, dj3d_extra=None, norm_from_acc=False, fused_counts=None):
        """Backward of the key-point objective down to dR (and the camera gradient).  ``fused_counts`` (device int64[V],
        samples per view of this batch): the forward of project_and_loss runs in the same launch (nemo_kp_fwd_bwd) -- the
        caller then skips project_and_loss and only finalises the loss scalar."""
        L, st, ctx = self.lib, _stream(), self.ctx
        nq72 = max(ctx.nq * 72, 1)
        lt = LOSS_TYPES[self.args.loss]
        add_trans = 0 if self.start_global_traj_anywhere else 1
        if fused_counts is not None:
            assert dj3d_extra is None and not cams_only
            check(L.nemo_kp_fwd_bwd(
                ctx.handle, N, self.V, self.T, dptr(w['A']), dptr(w['Jp']), dptr(Mq), nq72, dptr(w['TR']), HEAD_LD, add_trans,
                dptr(view_idx), dptr(frame_idx), self.p('learned_cameras'), dptr(self.targets), dptr(self.gt_size),
                FOCAL_LENGTH, self.cx, self.cy, lt, mean_mode, fused_counts, upstream, dptr(w['j3d']), dptr(w['p2d']),
                dptr(w['loss_all']), dptr(w['view_acc']), dptr(w['dA']), dptr(w['dJp']), dptr(w['dMq']), dptr(w['dTR']),
                HEAD_LD, self.g('learned_cameras'), self.nvalid, st), 'nemo_kp_fwd_bwd')
            # the per-view accumulators are complete here: finalize_kp (on another stream) need not wait for the rest of this chain
            self.kp_acc_done = torch.cuda.current_stream().record_event()
        else:
            check(L.nemo_kp_bwd_ex(
                ctx.handle, N, self.V, self.T, dptr(w['A']), dptr(w['Jp']), dptr(Mq), nq72, dptr(w['TR']),
                HEAD_LD, add_trans, dptr(view_idx), dptr(frame_idx), self.p('learned_cameras'), dptr(self.targets),
                dptr(self.gt_size), FOCAL_LENGTH, self.cx, self.cy, lt, mean_mode, dptr(w['view_acc']),
                None if norm_from_acc else dptr(w['norm']), upstream, None if cams_only else dptr(w['dA']),
                None if cams_only else dptr(w['dJp']), None if cams_only else dptr(w['dMq']),
                None if cams_only else dptr(w['dTR']), HEAD_LD, self.g('learned_cameras'), dptr(dj3d_extra), self.nvalid, st),
                'nemo_kp_bwd_ex')
        if cams_only:
            return
        if self.detach_articulation:
            w['dR'].zero_()
            return
        dPF = None
        if ctx.nq and not detach_pose:
            self.gemm(0, 1, N, 207, ctx.nq * 72, dptr(w['dMq']), nq72, ctx.C1, ctx.nq * 72, dptr(w['dPF']),
                      208)
            dPF = dptr(w['dPF'])
        check(L.nemo_fk_bwd(ctx.handle, N, dptr(w['R']), dptr(w['A']), dptr(w['dA']), dptr(w['dJp']), dPF,
                            208, dptr(w['dR']), st), 'nemo_fk_bwd')
        if detach_pose:
            w['dR'][:, 1:].zero_()       # body rotmats detached, global orient keeps its gradient (:4031)

    def backward_vposer_kl(self, w, N, weight):
        """d(weight*KL)/d poses[:, :63] through the frozen encoder, accumulated into dAA[:, 3:66]."""
        vp = self.vp
        if self._use_xp_vp(w):
            vx, m, vm = self._vpx, (lambda k: self._xm(w, k)), self._vm
            if self.xp_fmt == 2:
                self.absmax_xp([(dptr(w['dMULV']), N, 64, 64, m('vdM'))], overwrite=True)
            self.cast_xp([(dptr(w['dMULV']), N, 64, 64, w['dMULVx'], None, m('vdM'))])
            self.gemm_xp(N, 512, 64, w['dMULVx'], vx['emwT'], alpha=weight, maskx=w['E1x'], mask_mode=2, Cx=w['dE_ax'],
                         mA=m('vdM'), mB=vm('emw'), mOut=m('vdE'))
            self.gemm_xp(N, 63, 512, w['dE_ax'], vx['e2wT'], w['dAA'].data_ptr() + 4 * 3, 72, out_mode=1, mA=m('vdE'), mB=vm('e2w'))
            return
        if self.b16mem:
            vb = self.vpb
            self._cast(N, 64, dptr(w['dMULV']), 64, w['dMULVb'], 0)
            self.gemm16(N, 512, 64, w['dMULVb'], vb['emwT'], dptr(w['dE_a']), 512, alpha=weight, mask=dptr(w['E1']), ldmask=512,
                        mask_mode=2, Cb=w['dE_ab'])
            self.gemm16(N, 63, 512, w['dE_ab'], vb['e2wT'], w['dAA'].data_ptr() + 4 * 3, 72, out_mode=1)
            return
        self.gemm(0, 0, N, 512, 64, dptr(w['dMULV']), 64, dptr(vp['emw']), 512, dptr(w['dE_a']), 512,
                  alpha=weight, mask=dptr(w['E1']), ldmask=512, mask_mode=2, dense=True)
        self.gemm(0, 0, N, 63, 512, dptr(w['dE_a']), 512, dptr(vp['e2w_p']), 64,
                  w['dAA'].data_ptr() + 4 * 3, 72, out_mode=1, dense=True)

    def backward_mlp(self, w, N, view_idx, frame_idx, raw_phase, has_trans_grad=True, stages=(0, 1, 2), bucketed=False):
        """dROT (N+1,144), dTR (N+1,3) -> all MLP / RBF / phase / code gradients.

        ``stages``: the backward in the order it completes parameter gradients -- 0: heads + layer 4 (and the dX GEMMs
        down to layer 4's input gradient), 1: layer 2, 2: layer 0, RBF widths, instance codes, phase networks.
        ``bucketed``: the bias column sums of a stage are flushed and the side stream is joined at the END OF EVERY STAGE
        (each stage can then be a launch of its own with a gradient all-reduce behind it, dist.py); otherwise one batched
        column-sum launch at the very end, as the single-GPU step wants it."""
        L, st, h, r = self.lib, _stream(), self.h, N + 1
        if self.version == 0:
            return self._backward_nets_v0(w, N, view_idx, frame_idx, raw_phase, has_trans_grad)
        lm = 'learned_motion.'
        # merged heads: dHEAD (N+1, [rot6d 144 | trans 3]) against [rot_out.weight ; linear_out.weight].
        # trans - trans_0 cancels the linear_out bias exactly (the reference's autograd also yields an
        # exact 0 there), so its column sum is skipped unless the global trajectory is un-anchored.
        nout = 147 if has_trans_grad else 144
        nbias = 147 if (has_trans_grad and self.start_global_traj_anywhere) else 144
        if self.b16mem:
            return self._backward_mlp_b16(w, N, view_idx, frame_idx, raw_phase, nout, nbias, tuple(stages), bucketed)
        if self._use_xp(w):
            return self._backward_mlp_xp(w, N, view_idx, frame_idx, raw_phase, nout, nbias, tuple(stages), bucketed)
        # Schedule.  The parameter-gradient GEMMs (dW) are off the dependency chain: up to DW_ASIDE_ROWS rows they ALL go to
        # the side stream, each as soon as its dY exists, and the chain dX_head -> dX4 -> dX2 -> dX0 -> phase backward runs
        # uninterrupted on the main stream.  Beyond that (C4: every hidden-layer GEMM is thousands of tiles) only the heads'
        # dW goes beside the chain; dX and the hidden dW alternate on the main stream up to layer 2, then one fork: the (small)
        # layer-0 dW GEMM and the batched bias column sums on the side stream, the layer-0 dX GEMM and the three phase / RBF /
        # code backward kernels that consume it on the main stream.
        main, side = torch.cuda.current_stream(), self.side_stream
        cs_in = not bucketed      # bias column sums inside the phase backward's launch
        small = r <= self.SMALL_BATCH_ROWS
        # Round 5 (fp32 arithmetic only): the three activation-gradient launches leave the per-band column sums of their result
        # (nemo_gemm_f32_colsum): the bias gradients of layers 4, 2, 0 are sums over R = 2 ceil(r / 64) short rows, not passes
        # over r x h matrices (at C4 a 6.7 GB read per step; at 8 x 300 the tail kernel of the step).
        ecs = not self.bf16
        R = int(L.nemo_gemm_colsum_rows(r))
        cs4, cs2, cs0 = (w['cs4'], w['cs2'], w['cs0']) if ecs else (None, None, None)

        def bias_from(cs_buf, name):
            """the bias gradient of layer `name` as the column sum of the band sums; returns the gb to hand to the dW call"""
            if not ecs:
                return self.g(lm + name)
            self._colsums.append((dptr(cs_buf), R, h, h, self.g(lm + name)))
            return None
        # Round 4: the parameter-gradient products go beside the dX chain at (nearly) every size of an un-bucketed step -- same
        # box, ms per step, all of them beside / only the heads' beside / none (the round-2 schedule for > 1024 rows): 2400 rows
        # 1.386 / 1.380 / 1.405, 12 000: 5.581 / 5.607 / 5.718, 19 200: 8.766 / 8.864 / -, 38 400: 16.98 / 17.07 / -, 262 144 (C4):
        # 116.5 / 114.1 / 114.2.  Round 2 measured "no gain" for the 1000 x 1000 ones on the kernels of that round.
        aside = small or (not bucketed and r <= self.DW_ASIDE_ROWS)

        # (Enqueue order in small mode: the chain's next GEMM BEFORE the parameter-gradient GEMM that branches off -- a
        #  replayed graph keeps a node's first successor on its hardware queue and pays a 15 - 30 us cross-queue barrier
        #  for the others; that must not be the chain.)
        def dW(ready, *a, **k):
            if ready is None:
                return self._linear_bwd_params(*a, **k)
            side.wait_event(ready)
            with torch.cuda.stream(side):
                self._linear_bwd_params(*a, **k)

        def dY_ready(heads=False):
            # (the two small products of the output heads go beside the chain at every size of an un-bucketed step)
            return main.record_event() if (aside or (heads and not bucketed)) else None

        # Round 3: ALL parameter-gradient GEMMs of the backward as ONE grouped launch (nemo_gemm_grouped_f32) on the side
        # stream once the last dY of the dX chain exists, beside the layer-0 dX GEMM and the phase backward on the main
        # stream: four launches of 32 - 256 tiles (each with its own pipeline fill / output burst at ~one block per CU) ->
        # one of 592 tiles.  Measured (same box, un-profiled): one-instance shard 0.493 ms grouped against 0.501 per layer;
        # 8 x 300: 1.567 against 1.559 (there the per-layer launches on the 64 x 64 / 8-wave skinny configuration are ahead)
        # -> small batches only.  Round 4, random minibatches of 512 (the published run's mode): the per-layer launches, each on
        # the side stream as soon as its dY exists, are ahead again (0.575 against 0.590 ms per step; level at 600 rows, the
        # grouped launch 0.478 against 0.481 at 300) -> grouped up to GROUPED_DW_ROWS rows.
        if not bucketed and tuple(stages) == (0, 1, 2) and r <= self.GROUPED_DW_ROWS:
            dWs = []

            def dWg(rows, x, ldx_, fin, dy, lddy, fout, gw, gb, nbias=None):
                dWs.append((1, 0, fout, fin, rows, dy, lddy, x, ldx_, gw, fin, 1))
                if gb is not None:
                    self._colsums.append((dy, rows, fout if nbias is None else nbias, lddy, gb))
            dWg(r, dptr(w['H3']), h, h, dptr(w['dHEAD']), HEAD_LD, nout, self.g(lm + 'rot_out.weight'),
                self.g(lm + 'rot_out.bias'), nbias=nbias)
            self.gemm(0, 0, r, h, nout, dptr(w['dHEAD']), HEAD_LD, self.p(lm + 'rot_out.weight'), h,
                      dptr(w['dH']), h, mask=dptr(w['H3']), ldmask=h, mask_mode=1, dense=True, colsum=cs4)
            dWg(r, dptr(w['H2']), h, h, dptr(w['dH']), h, h, self.g(lm + 'net.net.4.weight'), bias_from(cs4, 'net.net.4.bias'))
            self.gemm(0, 0, r, h, h, dptr(w['dH']), h, self.p(lm + 'net.net.4.weight'), h, dptr(w['dH_b']), h,
                      mask=dptr(w['H2']), ldmask=h, mask_mode=1, tag='gemm_mlp_hidden_dx', dense=True, colsum=cs2)
            dWg(r, dptr(w['H1']), h, h, dptr(w['dH_b']), h, h, self.g(lm + 'net.net.2.weight'), bias_from(cs2, 'net.net.2.bias'))
            self.gemm(0, 0, r, h, h, dptr(w['dH_b']), h, self.p(lm + 'net.net.2.weight'), h, dptr(w['dH_c']), h,
                      mask=dptr(w['H1']), ldmask=h, mask_mode=1, dense=True, colsum=cs0)
            dWg(r, dptr(w['X']), self.ldx, self.din, dptr(w['dH_c']), h, h, self.g(lm + 'net.net.0.weight'),
                bias_from(cs0, 'net.net.0.bias'))
            ev = main.record_event()
            self.gemm(0, 0, r, self.din, h, dptr(w['dH_c']), h, self.p(lm + 'net.net.0.weight'), self.din,
                      dptr(w['dX']), self.ldx, dense=True)
            self.phase_bwd(w, N, view_idx, frame_idx, raw_phase, with_colsums=cs_in)
            side.wait_event(ev)
            with torch.cuda.stream(side):
                self.gemm_grouped(dWs)
                self.flush_colsums()
            main.wait_stream(side)
            return

        def end_of_stage():
            """bucketed: this stage's parameter gradients (incl. the bias column sums) are complete on `main`."""
            if not bucketed:
                return
            if small:
                with torch.cuda.stream(side):
                    self.flush_colsums()
                main.wait_stream(side)
            else:
                self.flush_colsums()

        if 0 in stages:
            ev = dY_ready(heads=True)
            a_ = (r, dptr(w['H3']), h, h, dptr(w['dHEAD']), HEAD_LD, nout, self.g(lm + 'rot_out.weight'), self.g(lm + 'rot_out.bias'))
            if ev is None:
                dW(None, *a_, nbias=nbias)
            self.gemm(0, 0, r, h, nout, dptr(w['dHEAD']), HEAD_LD, self.p(lm + 'rot_out.weight'), h,
                      dptr(w['dH']), h, mask=dptr(w['H3']), ldmask=h, mask_mode=1, dense=True, colsum=cs4)
            if ev is not None:
                dW(ev, *a_, nbias=nbias)
            ev = dY_ready()
            a_ = (r, dptr(w['H2']), h, h, dptr(w['dH']), h, h, self.g(lm + 'net.net.4.weight'), bias_from(cs4, 'net.net.4.bias'))
            if ev is None:
                dW(None, *a_)
            self.gemm(0, 0, r, h, h, dptr(w['dH']), h, self.p(lm + 'net.net.4.weight'), h, dptr(w['dH_b']), h,
                      mask=dptr(w['H2']), ldmask=h, mask_mode=1, tag='gemm_mlp_hidden_dx', dense=True, colsum=cs2)
            if ev is not None:
                dW(ev, *a_)
            end_of_stage()
        if 1 in stages:
            ev = dY_ready()
            a_ = (r, dptr(w['H1']), h, h, dptr(w['dH_b']), h, h, self.g(lm + 'net.net.2.weight'), bias_from(cs2, 'net.net.2.bias'))
            if ev is None:
                dW(None, *a_)
            self.gemm(0, 0, r, h, h, dptr(w['dH_b']), h, self.p(lm + 'net.net.2.weight'), h, dptr(w['dH_c']), h,
                      mask=dptr(w['H1']), ldmask=h, mask_mode=1, dense=True, colsum=cs0)
            if ev is not None:
                dW(ev, *a_)
            end_of_stage()
        if 2 not in stages:
            return

        def phase_bwd():
            self.phase_bwd(w, N, view_idx, frame_idx, raw_phase, with_colsums=cs_in)

        def dX0():
            self.gemm(0, 0, r, self.din, h, dptr(w['dH_c']), h, self.p(lm + 'net.net.0.weight'), self.din,
                      dptr(w['dX']), self.ldx, dense=True)
        w0 = (r, dptr(w['X']), self.ldx, self.din, dptr(w['dH_c']), h, h,
              self.g(lm + 'net.net.0.weight'), bias_from(cs0, 'net.net.0.bias'))
        # ONE more fork, one join: the side stream takes the layer-0 parameter gradient and the batched bias column
        # sums (every dY exists from here on), the main stream -- enqueued first, see above -- the layer-0 dX GEMM and
        # the three phase / RBF / code kernels that consume it
        ev = main.record_event()
        dX0()
        if cs_in and w0[8] is not None:   # (the layer-0 bias column sum rides in the phase launch with the others)
            self._colsums.append((w0[4], w0[0], w0[6], w0[5], w0[8]))
            w0 = w0[:8] + (None,)
        phase_bwd()
        side.wait_event(ev)
        with torch.cuda.stream(side):
            self._linear_bwd_params(*w0)
            self.flush_colsums()
        main.wait_stream(side)

    def finish_trans_grad(self, w, N):
        """d trans_0 = - sum_s d trans_s  (row N of dTR), :3764-3766."""
        if self.start_global_traj_anywhere:
            w['dTR'][N].zero_()
        else:
            check(self.lib.nemo_scale_neg_rowsum(N, 3, dptr(w['dTR']), HEAD_LD,
                                                 w['dTR'].data_ptr() + 4 * HEAD_LD * N, _stream()),
                  'nemo_scale_neg_rowsum')

    # ------------------------------------------------------------------ optimiser
    def _fill_segs(self, arr, segments):
        for i, s in enumerate(segments):
            t = s['step']
            arr[i].offset, arr[i].numel = s['offset'], s['numel']
            arr[i].lr, arr[i].weight_decay = s['lr'], s['wd']
            arr[i].step_size = s['lr'] / (1.0 - 0.9 ** t)
            arr[i].bias_corr2_sqrt = math.sqrt(1.0 - 0.999 ** t)
            arr[i].adamw = 1 if s['adamw'] else 0

    def adam(self, segments, exp_avg=None, exp_avg_sq=None):
        """segments: list of dicts(offset, numel, lr, wd, adamw, step) -- step is the NEW step count."""
        if not segments:
            return
        self._wrec_ok = False
        arr = (AdamSeg * len(segments))()
        self._fill_segs(arr, segments)
        m = self.exp_avg if exp_avg is None else exp_avg
        v = self.exp_avg_sq if exp_avg_sq is None else exp_avg_sq
        for i in range(0, len(segments), _lib.ADAM_MAX_SEG):
            n = min(_lib.ADAM_MAX_SEG, len(segments) - i)
            check(self.lib.nemo_adam_step(n, ctypes.cast(ctypes.byref(arr, i * ctypes.sizeof(AdamSeg)),
                                                         ctypes.POINTER(AdamSeg)),
                                          self.params.data_ptr(), self.grads.data_ptr(), m.data_ptr(),
                                          v.data_ptr(), 0.9, 0.999, 1e-8, _stream()), 'nemo_adam_step')

    # ---- graph-capturable variant: the segment table lives in device memory and is advanced ON the device
    def adam_table_sync(self, segments):
        """Make the device-resident segment table describe the update BEFORE this step's (step counts t - 1):
        ``step_begin`` inside the captured step advances it to t and refreshes the bias corrections, so in the steady
        state nothing is uploaded.  A (pinned, async) upload happens only when a learning rate, the segment list or
        the step counts differ from what the device holds -- first use, a plateau-scheduler drop, a loaded optimiser
        state, steps taken meanwhile by another path (warm-up, camera fit, eager steps)."""
        n = len(segments)
        assert 0 < n <= _lib.ADAM_MAX_SEG
        if self._seg_host is None:
            nbytes = ctypes.sizeof(AdamSeg) * _lib.ADAM_MAX_SEG
            self._seg_host = torch.zeros(nbytes, dtype=torch.uint8).pin_memory()
            self._seg_dev = torch.zeros(nbytes, dtype=torch.uint8, device=self.device)
            self._seg_shadow = None
        want = [(s_['offset'], s_['numel'], s_['lr'], s_['wd'], bool(s_['adamw']), s_['step'] - 1) for s_ in segments]
        if want != self._seg_shadow:
            arr = (AdamSeg * n).from_address(self._seg_host.data_ptr())
            for i, (off, numel, lr, wd, adamw, t_prev) in enumerate(want):
                arr[i].offset, arr[i].numel, arr[i].lr, arr[i].weight_decay = off, numel, lr, wd
                arr[i].adamw, arr[i].step = 1 if adamw else 0, t_prev
                arr[i].step_size, arr[i].bias_corr2_sqrt = 0.0, 1.0          # (written by step_begin)
            self._seg_dev.copy_(self._seg_host, non_blocking=True)
            self._seg_shadow = want
        # what the table holds once this step's step_begin has run: the caller confirms with adam_table_commit() AFTER the
        # launch / replay has been enqueued (a body that raises in between leaves the shadow describing what the device
        # really holds, and the next step re-uploads)
        self._seg_pending = [(off, numel, lr, wd, adamw, t_prev + 1) for off, numel, lr, wd, adamw, t_prev in want]
        return n, max(s_['numel'] for s_ in segments)

    def adam_table_commit(self):
        self._seg_shadow, self._seg_pending = self._seg_pending, None

    def adam_table_invalidate(self):
        """The device table may have been advanced by a launch the host did not see complete: upload it next time."""
        self._seg_shadow = None

    def step_begin(self, arena, grads=None, n_seg=0):
        """First launch of an iteration whose first kernel is not the phase kernel (camera fit): zero the workspace's
        accumulator arena and `grads` (a slice of the flat gradient buffer, or None) and advance the device Adam table --
        one kernel instead of two memsets and a copy."""
        self.bind_reduce_ws()
        check(self.lib.nemo_step_begin(arena.data_ptr(), arena.numel() * 4, dptr(grads),
                                       grads.numel() * 4 if grads is not None else 0,
                                       self._seg_dev.data_ptr() if n_seg else None, n_seg, 0.9, 0.999, _stream()),
              'nemo_step_begin')

    def adam_from_table(self, n, max_numel, start=0, exp_avg=None, exp_avg_sq=None, guard=None):
        """Fused Adam over segments [start, start + n) of the device-resident table; ``guard``: a device scalar -- no
        update when it is non-zero (warm-up: NaN gradients were counted)."""
        m = self.exp_avg if exp_avg is None else exp_avg
        v = self.exp_avg_sq if exp_avg_sq is None else exp_avg_sq
        self._wrec_ok = False
        check(self.lib.nemo_adam_step_dev_if(n, self._seg_dev.data_ptr() + start * ctypes.sizeof(AdamSeg), max_numel,
                                             self.params.data_ptr(), self.grads.data_ptr(), m.data_ptr(), v.data_ptr(),
                                             0.9, 0.999, 1e-8, dptr(guard), _stream()), 'nemo_adam_step_dev_if')

    def sequence(self, steps, B, view_idx=None, frame_idx=None):
        """Device-side state of a fit phase the host does not watch iteration by iteration (warm-up, camera fit): the
        phase's (steps, B) index tables (drawn up front), a (steps, 8) loss log, the iteration counter that selects the
        rows (nemo_seq_gather / nemo_seq_log) and a sticky NaN-gradient count.  The buffers are PERSISTENT per batch size
        (captured graphs of an iteration hold their addresses); they only grow, and `gen` -- part of the graph keys --
        changes when they do."""
        if not hasattr(self, '_seq'):
            self._seq, self._seq_gen = {}, 0
        seq = self._seq.get(B)
        if seq is None or seq['cap'] < steps:
            dev, cap = self.device, max(steps, 1024)
            self._seq_gen += 1
            seq = self._seq[B] = {'cap': cap, 'gen': self._seq_gen, 'log': torch.zeros(cap, 8, dtype=torch.float32, device=dev),
                                  'counter': torch.zeros(1, dtype=torch.int32, device=dev),
                                  'nan': torch.zeros(1, dtype=torch.float32, device=dev)}
            if B:
                seq['vi'] = torch.zeros(cap, B, dtype=torch.long, device=dev)
                seq['fi'] = torch.zeros(cap, B, dtype=torch.long, device=dev)
        seq['counter'].zero_()
        seq['nan'].zero_()
        if B:
            seq['vi'][:steps].copy_(view_idx.to(torch.long).contiguous().pin_memory(), non_blocking=True)
            seq['fi'][:steps].copy_(frame_idx.to(torch.long).contiguous().pin_memory(), non_blocking=True)
        return seq

    def scratch_moments(self):
        """Zeroed Adam moment buffers of a throw-away optimiser (opt_cam builds a fresh Adam on the cameras, :2870):
        persistent storage, because the captured iteration holds their addresses."""
        if not hasattr(self, '_tmp_m'):
            self._tmp_m, self._tmp_v = torch.zeros_like(self.exp_avg), torch.zeros_like(self.exp_avg_sq)
        self._tmp_m.zero_()
        self._tmp_v.zero_()
        return self._tmp_m, self._tmp_v

    def read_scalars(self):
        self._scal_host.copy_(self.scal, non_blocking=True)
        torch.cuda.current_stream().synchronize()
        return self._scal_host.numpy().copy()

    def publish_scalars(self, src=None):
        """Enqueue (or capture) the copy of the loss scalars to pinned host memory + flag; everything
        enqueued afterwards (MLP backward, Adam) keeps running while the host reads them."""
        check(self.lib.nemo_publish_scalars((self.scal if src is None else src).data_ptr(), 8, self._pub_host.data_ptr(),
                                            self._pub_host.data_ptr() + 32, _stream()), 'nemo_publish_scalars')

    def arm_scalars(self):
        """Host side, BEFORE the launch that contains publish_scalars (no publish may be in flight)."""
        self._pub_flag[0] = 0

    def wait_scalars(self, timeout_s=120.0):
        """Poll the pinned flag.  Steps of the sizes this engine is built for last 0.3 - 2 ms and the next launch is
        latency-critical, so the thread spins for the first 2 ms (measured: yielding the core between polls --
        ``sleep(0)`` -- makes the step time jitter by 3 - 8 %; the clock is read every 256th poll only); a wait that
        outlasts that (a C4-size step takes > 100 ms) sleeps 50 us between polls so that one rank per GPU does not pin
        eight host cores."""
        flag = self._pub_flag
        t0, polls = time.monotonic(), 0
        while flag[0] == 0:
            polls += 1
            if polls & 0xFF == 0 and time.monotonic() - t0 > 2e-3:
                break
        while flag[0] == 0:
            time.sleep(5e-5)
            if time.monotonic() - t0 > timeout_s:
                raise _lib.NemoHipError('loss read-back flag never raised (device fault or hung kernel?)')
        return self._pub_np[:8].copy()
