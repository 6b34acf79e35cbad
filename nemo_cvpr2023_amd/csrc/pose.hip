// Per-sample pose kernels: phase warp + RBF embedding, rot6d -> R -> axis-angle, Rodrigues.
// All of these are tiny (O(100) flops per sample-joint); they are written one thread per
// (sample, joint) / per (sample), with coalesced reads along the (instance x frame) batch axis,
// and exist to replace ~100 aten launches per step of the reference by 1 each.
#include <cstdlib>
#include <type_traits>
#include "common.h"
#include "rot_math.h"
#include "absmax.h"
#include "../../include/nemo_hip.h"

extern "C" int32_t nemo_abi_version(void) { return NEMO_ABI_VERSION; }

namespace {

// ------------------------------------------------------------------------------------------
// RBF kernels, nemo/rbf.py:62-120.  a = (x-c)^2 / exp(log_sigma).
__device__ __forceinline__ float rbf_phi(int id, float a) {
    switch (id) {
        case 0: return a * a;                                   // quadratic
        case 1: return a;                                       // linear
        case 2: return expf(-(a * a));                          // gaussian
        case 3: return 1.f / (1.f + a * a);                     // inverse_quadratic
        case 4: return sqrtf(1.f + a * a);                      // multiquadric
        case 5: return 1.f / sqrtf(1.f + a * a);                // inverse_multiquadric
        case 6: return a * a * logf(a + 1.f);                   // spline
        case 7: return (a - 1.f) * expf(-a);                    // poisson_one
        case 8: return ((a - 2.f) / 2.f) * a * expf(-a);        // poisson_two
        case 9: return (1.f + 1.7320508075688772f * a) * expf(-1.7320508075688772f * a);   // matern32
        default: return (1.f + 2.23606797749979f * a + (5.f / 3.f) * a * a) * expf(-2.23606797749979f * a);
    }
}
__device__ __forceinline__ float rbf_dphi(int id, float a) {
    switch (id) {
        case 0: return 2.f * a;
        case 1: return 1.f;
        case 2: return -2.f * a * expf(-(a * a));
        case 3: { const float u = 1.f + a * a; return -2.f * a / (u * u); }
        case 4: return a / sqrtf(1.f + a * a);
        case 5: { const float u = 1.f + a * a; return -a / (u * sqrtf(u)); }
        case 6: return 2.f * a * logf(a + 1.f) + a * a / (a + 1.f);
        case 7: return expf(-a) * (2.f - a);
        case 8: return expf(-a) * (-0.5f * a * a + 2.f * a - 1.f);
        case 9: return -3.f * a * expf(-1.7320508075688772f * a);
        default: { const float r5 = 2.23606797749979f;
                   return expf(-r5 * a) * (-(5.f / 3.f) * a - (5.f * r5 / 3.f) * a * a); }
    }
}

__device__ __forceinline__ float sigmoidf_(float z) { return 1.f / (1.f + expf(-z)); }

__device__ __forceinline__ float lin01(long i, long n) {
    // torch.linspace(0, 1, n)[i]: start + i*step below the midpoint, end - (n-1-i)*step above
    if (n <= 1) return 0.f;
    const float step = 1.0f / (float)(n - 1);
    return (i < n / 2) ? (float)i * step : 1.0f - (float)(n - 1 - i) * step;
}

// One wave per sample.  Lanes stride over the K nodes of the owning phase network (the three passes
// x, 0, 1 of monotonic_network.py:23-39 share one sweep), wave-reduce, then stride over the D RBF
// centres / C code entries of the MLP input row.
struct PhaseFwdArgs {
    long N, V, T;
    int K, D, C;
    const int64_t* view_idx; const int64_t* frame_idx; const float* raw_phase;
    const float* shifts; const float* scales; long ldp;
    const float* log_sigmas; const float* codes; const float* code_noise;
    int kid; float* X; long ldx; float* phase_out; float* den_out;
    float* x_meta;      // may be NULL: scale record of X (include/nemo_hip.h nemo_gemm_xp fmt 2) -- the rows' absmax is accumulated into it
};
// returns max |value| this lane wrote into X (0 for a wave without a sample)
__device__ __forceinline__ float phase_embed_fwd_body(const PhaseFwdArgs& a, long bid) {
    const long N = a.N, T = a.T, ldp = a.ldp, ldx = a.ldx;
    const int K = a.K, D = a.D, C = a.C, kid = a.kid;
    const int64_t* __restrict__ view_idx = a.view_idx;
    const int64_t* __restrict__ frame_idx = a.frame_idx;
    const float* __restrict__ raw_phase = a.raw_phase;
    const float* __restrict__ shifts = a.shifts;
    const float* __restrict__ scales = a.scales;
    const float* __restrict__ log_sigmas = a.log_sigmas;
    const float* __restrict__ codes = a.codes;
    const float* __restrict__ code_noise = a.code_noise;
    float* __restrict__ X = a.X;
    float* __restrict__ phase_out = a.phase_out;
    float* __restrict__ den_out = a.den_out;
    const long s = bid * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (s > N) return 0.f;                  // wave-uniform
    float ph = 0.f;
    long v = 0;
    if (s < N) {
        v = view_idx[s];
        const float x = raw_phase ? raw_phase[s] : lin01(frame_idx[s], T);
        const float* sh = shifts + v * ldp;
        const float* sc = scales + v * ldp;
        float y = 0.f, z = 0.f, o = 0.f;
        for (int k = lane; k < K; k += 64) {
            const float shp = fmaxf(sh[k], 0.f), scp = fmaxf(sc[k], 0.f);
            y += sigmoidf_(scp * (x - shp));
            z += sigmoidf_(scp * (0.f - shp));
            o += sigmoidf_(scp * (1.f - shp));
        }
        y = wave_sum(y) / (float)K; z = wave_sum(z) / (float)K; o = wave_sum(o) / (float)K;
        ph = (y - z) / (o - z + 1e-6f);                 // monotonic_network.py:33-39
        if (phase_out && lane == 0) phase_out[s] = ph;
        if (den_out && lane == 0) den_out[s] = o - z + 1e-6f;
    }
    float* xr = X + s * ldx;
    float mx = 0.f;
    if (D > 0) {
        for (int d = lane; d < D; d += 64) {
            const float diff = ph - lin01(d, D);        // centres = linspace(0,1,D)  rbf.py:38-39
            const float u = rbf_phi(kid, (diff * diff) / expf(log_sigmas[d]));
            xr[d] = u;
            mx = fmaxf(mx, fabsf(u));
        }
    } else if (lane == 0) {
        xr[0] = ph;
        mx = fabsf(ph);
    }
    const int off = D > 0 ? D : 1;
    for (int c = lane; c < C; c += 64) {
        float cv = 0.f;
        if (s < N) {
            cv = codes[v * C + c];
            if (code_noise) cv += code_noise[s * C + c];
        }
        xr[off + c] = cv;
        mx = fmaxf(mx, fabsf(cv));
    }
    return mx;
}
// the block's max |X| into X's scale record: ONE atomic per block (one per wave -- 2401 on 32 slots -- took the launch from 10 to 24 us)
__device__ __forceinline__ void phase_embed_put_absmax(float* x_meta, float mx) {
    __shared__ float wmx[4];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
    if ((threadIdx.x & 63) == 0) wmx[threadIdx.x >> 6] = mx;
    __syncthreads();
    if (threadIdx.x == 0) nemo_meta::meta_absmax_put(x_meta, (int)blockIdx.x, fmaxf(fmaxf(wmx[0], wmx[1]), fmaxf(wmx[2], wmx[3])));
}
__global__ __launch_bounds__(256) void phase_embed_fwd_kernel(PhaseFwdArgs a) {
    const float mx = phase_embed_fwd_body(a, (long)blockIdx.x);
    if (a.x_meta) phase_embed_put_absmax(a.x_meta, mx);
}

// The FIRST launch of a step: phase / RBF / code forward in blocks [0, nb_phase), and in the blocks behind them what
// nemo_step_begin does (zero-fill of the gradient buffer and of the workspace's accumulator arena -- nothing the phase blocks
// touch --, the per-update bookkeeping of the device-resident Adam table): one graph node less on the step's critical chain.
struct StepBeginArgs {
    float4* z0; long n0; float4* z1; long n1; float* t0; int r0; float* t1; int r1;
    nemo_adam_seg* segs; int n_seg; double b1, b2;
};
// (round 6, last part) ... and the absmax pass over the split-precision chain's weights (nemo_absmax_multi's blocks): the LEADING nb_abs
// blocks of the grid -- the longest ones, dispatched first.  Beside the phase kernel on a second stream that pass cost a fork and a
// cross-queue join at the top of every update step (~38 us from the end of Adam to the chain's cast launch; ~20 in one launch).
__global__ __launch_bounds__(256) void phase_embed_begin_kernel(PhaseFwdArgs a, StepBeginArgs b, int nb_phase, int nb_abs, xp::AbsmaxArgs am) {
    if ((int)blockIdx.x < nb_abs) {                // (block-uniform)
        xp::absmax_block(am, (int)blockIdx.x, nb_abs);
        return;
    }
    const int bid = (int)blockIdx.x - nb_abs;
    if (bid < nb_phase) {
        const float mx = phase_embed_fwd_body(a, (long)bid);
        if (a.x_meta) phase_embed_put_absmax(a.x_meta, mx);
        return;
    }
    const long nbz = (long)gridDim.x - nb_abs - nb_phase, bz = (long)bid - nb_phase;
    const long stride = nbz * blockDim.x, i0 = bz * blockDim.x + threadIdx.x;
    const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
    for (long i = i0; i < b.n0; i += stride) b.z0[i] = z;
    for (long i = i0; i < b.n1; i += stride) b.z1[i] = z;
    if (i0 < b.r0) b.t0[i0] = 0.f;
    if (i0 < b.r1) b.t1[i0] = 0.f;
    if (b.segs && bz == 0 && (int)threadIdx.x < b.n_seg) {
        nemo_adam_seg sg = b.segs[threadIdx.x];
        sg.step += 1;
        sg.step_size = (float)((double)sg.lr / (1.0 - pow(b.b1, (double)sg.step)));
        sg.bias_corr2_sqrt = (float)sqrt(1.0 - pow(b.b2, (double)sg.step));
        b.segs[threadIdx.x] = sg;
    }
}

// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void rot6d_fwd_kernel(long total, int J, const float* __restrict__ rot6d,
                                                        long ld6, int zero_nan, float* __restrict__ R,
                                                        float* __restrict__ aa) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const long row = i / J;
    const int j = (int)(i % J);
    float x[6], Rm[9];
    const float* src = rot6d + row * ld6 + j * 6;
#pragma unroll
    for (int k = 0; k < 6; ++k) x[k] = src[k];
    rot6d_fwd(x, Rm);
    if (R) {
#pragma unroll
        for (int k = 0; k < 9; ++k) R[i * 9 + k] = Rm[k];
    }
    if (aa) {
        float a[3];
        rotmat_to_aa_fwd(Rm, zero_nan, a);
        aa[i * 3 + 0] = a[0]; aa[i * 3 + 1] = a[1]; aa[i * 3 + 2] = a[2];
    }
}

__global__ __launch_bounds__(256) void rot6d_bwd_kernel(long total, int J, const float* __restrict__ rot6d,
                                                        long ld6, int zero_nan, const float* __restrict__ dR,
                                                        const float* __restrict__ daa,
                                                        float* __restrict__ d_rot6d, long ldd) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const long row = i / J;
    const int j = (int)(i % J);
    float x[6], Rm[9], g[9];
    const float* src = rot6d + row * ld6 + j * 6;
#pragma unroll
    for (int k = 0; k < 6; ++k) x[k] = src[k];
#pragma unroll
    for (int k = 0; k < 9; ++k) g[k] = dR ? dR[i * 9 + k] : 0.f;
    if (daa) {
        rot6d_fwd(x, Rm);
        const float ga[3] = {daa[i * 3], daa[i * 3 + 1], daa[i * 3 + 2]};
        if (ga[0] != 0.f || ga[1] != 0.f || ga[2] != 0.f) rotmat_to_aa_bwd(Rm, zero_nan, ga, g);
    }
    float dx[6];
    rot6d_bwd(x, g, dx);
    float* dst = d_rot6d + row * ldd + j * 6;
#pragma unroll
    for (int k = 0; k < 6; ++k) dst[k] = dx[k];
}

__global__ __launch_bounds__(256) void rotmat_to_aa_kernel(long M, const float* __restrict__ R, int zero_nan,
                                                           float* __restrict__ aa) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= M) return;
    float Rm[9], a[3];
#pragma unroll
    for (int k = 0; k < 9; ++k) Rm[k] = R[i * 9 + k];
    rotmat_to_aa_fwd(Rm, zero_nan, a);
    aa[i * 3] = a[0]; aa[i * 3 + 1] = a[1]; aa[i * 3 + 2] = a[2];
}

__global__ __launch_bounds__(256) void rodrigues_fwd_kernel(long M, const float* __restrict__ th, int form,
                                                            float* __restrict__ R) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= M) return;
    const float t[3] = {th[i * 3], th[i * 3 + 1], th[i * 3 + 2]};
    float Rm[9];
    if (form == 0) rodrigues_fwd(t, Rm); else rodrigues_lbs_fwd(t, Rm);
#pragma unroll
    for (int k = 0; k < 9; ++k) R[i * 9 + k] = Rm[k];
}

__global__ __launch_bounds__(256) void rodrigues_bwd_kernel(long M, const float* __restrict__ th,
                                                            const float* __restrict__ dR,
                                                            float* __restrict__ dth) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= M) return;
    const float t[3] = {th[i * 3], th[i * 3 + 1], th[i * 3 + 2]};
    float G[9], d[3];
#pragma unroll
    for (int k = 0; k < 9; ++k) G[k] = dR[i * 9 + k];
    rodrigues_bwd(t, G, d);
    dth[i * 3] = d[0]; dth[i * 3 + 1] = d[1]; dth[i * 3 + 2] = d[2];
}

// rows<N : joint 0 = R[s][0];  joints 1..23 = Rodrigues(aa[s][3j .. 3j+2])
// rows>=N: joint 0 = R[s][0];  joints 1..21 = Rodrigues(aa_dec[s][3(j-1)..]); 22,23 from aa
// dec6d != NULL (nemo_v2v_prep_fwd_dec): the decoder's 6-D output (N, 21, 6; row stride lddec) instead of its axis-angle
// form -- the thread of (second body, joint 1..21) does the conversion 6-D -> R -> axis-angle itself (what a
// nemo_rot6d_fwd launch in front of this one did) and stores the axis-angle in aa_dec_out: one launch less on the chain
// the mesh kernel waits for.
__global__ __launch_bounds__(256) void v2v_prep_fwd_kernel(long N, const float* __restrict__ R,
                                                           const float* __restrict__ aa,
                                                           const float* __restrict__ aa_dec,
                                                           float* __restrict__ R2,
                                                           const int64_t* __restrict__ n_valid,
                                                           const float* __restrict__ dec6d = nullptr, long lddec = 0,
                                                           float* __restrict__ aa_dec_out = nullptr) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= 2 * N * 24) return;
    const long row = i / 24;
    const int j = (int)(i % 24);
    const long s = row < N ? row : row - N;
    // padding samples (s >= *n_valid): the second body repeats the first, so the full-mesh L1 term and its gradient
    // are exactly zero for them (same arithmetic on the same inputs; sign(0) = 0)
    const bool dec = !n_valid || s < *n_valid;
    float Rm[9];
    if (j == 0) {
#pragma unroll
        for (int k = 0; k < 9; ++k) Rm[k] = R[s * 216 + k];
    } else if (dec6d && row >= N && j <= 21) {
        float x[6], Rd[9], t[3];
        const float* src6 = dec6d + s * lddec + (j - 1) * 6;
#pragma unroll
        for (int k = 0; k < 6; ++k) x[k] = src6[k];
        rot6d_fwd(x, Rd);
        rotmat_to_aa_fwd(Rd, 0, t);
        aa_dec_out[s * 63 + (j - 1) * 3 + 0] = t[0];
        aa_dec_out[s * 63 + (j - 1) * 3 + 1] = t[1];
        aa_dec_out[s * 63 + (j - 1) * 3 + 2] = t[2];
        if (!dec) { t[0] = aa[s * 72 + j * 3]; t[1] = aa[s * 72 + j * 3 + 1]; t[2] = aa[s * 72 + j * 3 + 2]; }
        rodrigues_fwd(t, Rm);
    } else {
        const float* src = (row >= N && j <= 21 && dec) ? aa_dec + s * 63 + (j - 1) * 3 : aa + s * 72 + j * 3;
        const float t[3] = {src[0], src[1], src[2]};
        rodrigues_fwd(t, Rm);
    }
#pragma unroll
    for (int k = 0; k < 9; ++k) R2[i * 9 + k] = Rm[k];
}

__global__ __launch_bounds__(256) void v2v_prep_bwd_kernel(long N, const float* __restrict__ aa,
                                                           const float* __restrict__ dR2, float scale,
                                                           float* __restrict__ d_aa, float* __restrict__ dR) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N * 24) return;
    const long s = i / 24;
    const int j = (int)(i % 24);
    float G[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) G[k] = dR2[i * 9 + k];
    if (j == 0) {
#pragma unroll
        for (int k = 0; k < 9; ++k) dR[s * 216 + k] += scale * G[k];
    } else {
        const float t[3] = {aa[s * 72 + j * 3], aa[s * 72 + j * 3 + 1], aa[s * 72 + j * 3 + 2]};
        float d[3];
        rodrigues_bwd(t, G, d);
        d_aa[s * 72 + j * 3 + 0] += scale * d[0];
        d_aa[s * 72 + j * 3 + 1] += scale * d[1];
        d_aa[s * 72 + j * 3 + 2] += scale * d[2];
    }
}


// rot6d backward of the fit step with its two small neighbours folded in (three launches -> one):
//  * v2v_prep_bwd: the full-mesh term's gradient arrives as dR2 (per-joint rotation-matrix gradient of the
//    "orig" body); joint 0 adds it to dR, joints 1..23 pull it back through Rodrigues to axis-angle -- both
//    are consumed right here, so they are added in registers instead of round-tripping through dR / dAA;
//  * neg_rowsum: d trans_0 = - sum_s d trans_s (row N of dTR) is an independent reduction done by one extra
//    block at the end of the grid.
__global__ __launch_bounds__(256) void pose_bwd_fused_kernel(long N, const float* __restrict__ rot6d, long ld6,
                                                             int zero_nan, const float* __restrict__ dR,
                                                             const float* __restrict__ daa,
                                                             float* __restrict__ d_rot6d, long ldd,
                                                             const float* __restrict__ aa,
                                                             const float* __restrict__ dR2, float scale,
                                                             float* __restrict__ dTR, long ldt, int zero_row,
                                                             float* __restrict__ head_meta) {
    if (blockIdx.x == gridDim.x - 1) {                 // the reduction block
        // row N of the head gradient belongs to the "phase 0 / zero code" row of the MLP: its rotation columns carry no
        // gradient.  Workspaces are shared by batch sizes, so whatever an earlier, larger batch left there is cleared
        if (zero_row && threadIdx.x < 144) d_rot6d[N * ldd + threadIdx.x] = 0.f;
        if (!dTR) return;
        __shared__ float red[16];
        float mx = 0.f;
        for (int c = 0; c < 3; ++c) {
            float s = 0.f;
            for (long r = threadIdx.x; r < N; r += blockDim.x) {
                const float v = dTR[r * ldt + c];
                s += v;
                mx = fmaxf(mx, fabsf(v));
            }
            const float t = block_sum(s, red);
            if (threadIdx.x == 0) {
                dTR[N * ldt + c] = -t;
                mx = fmaxf(mx, fabsf(t));
            }
        }
        if (head_meta) {                               // the three translation columns of the head gradient, row N included
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
            if ((threadIdx.x & 63) == 0) nemo_meta::meta_absmax_put(head_meta, (int)(threadIdx.x >> 6), mx);
        }
        return;
    }
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    float hmx = 0.f;
    if (i < N * 24) {
    const long row = i / 24;
    const int j = (int)(i % 24);
    float x[6], Rm[9], g[9];
    const float* src = rot6d + row * ld6 + j * 6;
#pragma unroll
    for (int k = 0; k < 6; ++k) x[k] = src[k];
#pragma unroll
    for (int k = 0; k < 9; ++k) g[k] = dR ? dR[i * 9 + k] : 0.f;
    float ga[3] = {0.f, 0.f, 0.f};
    if (daa) { ga[0] = daa[i * 3]; ga[1] = daa[i * 3 + 1]; ga[2] = daa[i * 3 + 2]; }
    if (dR2) {
        float G[9];
#pragma unroll
        for (int k = 0; k < 9; ++k) G[k] = dR2[i * 9 + k];
        if (j == 0) {
#pragma unroll
            for (int k = 0; k < 9; ++k) g[k] += scale * G[k];
        } else {
            const float t[3] = {aa[row * 72 + j * 3], aa[row * 72 + j * 3 + 1], aa[row * 72 + j * 3 + 2]};
            float d[3];
            rodrigues_bwd(t, G, d);
            ga[0] += scale * d[0]; ga[1] += scale * d[1]; ga[2] += scale * d[2];
        }
    }
    if (ga[0] != 0.f || ga[1] != 0.f || ga[2] != 0.f) {
        rot6d_fwd(x, Rm);
        rotmat_to_aa_bwd(Rm, zero_nan, ga, g);
    }
    float dx[6];
    rot6d_bwd(x, g, dx);
    float* dst = d_rot6d + row * ldd + j * 6;
#pragma unroll
    for (int k = 0; k < 6; ++k) { dst[k] = dx[k]; hmx = fmaxf(hmx, fabsf(dx[k])); }
    }
    if (head_meta) {                                   // absmax of the 144 rotation columns written here: one atomic per block
        __shared__ float hw[4];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) hmx = fmaxf(hmx, __shfl_xor(hmx, o, 64));
        if ((threadIdx.x & 63) == 0) hw[threadIdx.x >> 6] = hmx;
        __syncthreads();
        if (threadIdx.x == 0) nemo_meta::meta_absmax_put(head_meta, (int)blockIdx.x, fmaxf(fmaxf(hw[0], hw[1]), fmaxf(hw[2], hw[3])));
    }
}

__global__ __launch_bounds__(1024) void neg_rowsum_kernel(long N, int cols, const float* __restrict__ X,
                                                          long ldx, float* __restrict__ out) {
    __shared__ float red[16];
    for (int c = 0; c < cols; ++c) {
        float s = 0.f;
        for (long r = threadIdx.x; r < N; r += blockDim.x) s += X[r * ldx + c];
        const float t = block_sum(s, red);
        if (threadIdx.x == 0) out[c] = -t;
    }
}

}  // namespace

#define GRID1D(n) dim3(nemo_cdiv((n), 256)), dim3(256), 0, (hipStream_t)stream

static int32_t phase_fwd_args(PhaseFwdArgs* a, int64_t N, int64_t V, int64_t T, int64_t K, int64_t D, int64_t C,
                              const int64_t* view_idx, const int64_t* frame_idx, const float* raw_phase, const float* shifts,
                              const float* scales, int64_t ldp, const float* log_sigmas, const float* codes,
                              const float* code_noise, int32_t kernel_id, float* X, int64_t ldx, float* phase_out,
                              float* den_out, float* x_meta) {
    if (N < 0 || V <= 0 || K <= 0 || D < 0 || C < 0 || !X || !shifts || !scales) return NEMO_EINVAL;
    if (N > 0 && (!view_idx || (!frame_idx && !raw_phase))) return NEMO_EINVAL;
    if ((D > 0 && !log_sigmas) || (C > 0 && !codes) || kernel_id < 0 || kernel_id > 10) return NEMO_EINVAL;
    if (ldx < (D > 0 ? D : 1) + C || ldp < K) return NEMO_EINVAL;
    *a = PhaseFwdArgs{(long)N, (long)V, (long)T, (int)K, (int)D, (int)C, view_idx, frame_idx, raw_phase, shifts, scales,
                      (long)ldp, log_sigmas, codes, code_noise, (int)kernel_id, X, (long)ldx, phase_out, den_out, x_meta};
    return NEMO_OK;
}

extern "C" int32_t nemo_phase_embed_fwd(int64_t N, int64_t V, int64_t T, int64_t K, int64_t D, int64_t C,
                                        const int64_t* view_idx, const int64_t* frame_idx,
                                        const float* raw_phase, const float* shifts, const float* scales,
                                        int64_t ldp, const float* log_sigmas, const float* codes,
                                        const float* code_noise, int32_t kernel_id, float* X, int64_t ldx,
                                        float* phase_out, float* den_out, float* x_meta, void* stream) {
    PhaseFwdArgs a;
    const int32_t rc = phase_fwd_args(&a, N, V, T, K, D, C, view_idx, frame_idx, raw_phase, shifts, scales, ldp, log_sigmas,
                                      codes, code_noise, kernel_id, X, ldx, phase_out, den_out, x_meta);
    if (rc) return rc;
    hipLaunchKernelGGL(phase_embed_fwd_kernel, dim3(nemo_cdiv(N + 1, 4)), dim3(256), 0, (hipStream_t)stream, a);
    NEMO_LAUNCH_CHECK();
    return NEMO_OK;
}

// nemo_phase_embed_fwd + nemo_step_begin (same arguments, same results) as ONE launch: the first node of a step's graph.
extern "C" int32_t nemo_phase_embed_fwd_begin(int64_t N, int64_t V, int64_t T, int64_t K, int64_t D, int64_t C,
                                              const int64_t* view_idx, const int64_t* frame_idx, const float* raw_phase,
                                              const float* shifts, const float* scales, int64_t ldp,
                                              const float* log_sigmas, const float* codes, const float* code_noise,
                                              int32_t kernel_id, float* X, int64_t ldx, float* phase_out, float* den_out,
                                              float* x_meta, void* z0, int64_t bytes0, void* z1, int64_t bytes1, nemo_adam_seg* segs_dev,
                                              int32_t n_seg, double beta1, double beta2, int32_t n_absmax,
                                              const nemo_absmax_desc* absmax, void* stream) {
    PhaseFwdArgs a;
    xp::AbsmaxArgs am;
    int nb_abs = 0;
    if (xp::absmax_fill(am, n_absmax, absmax, nb_abs)) return NEMO_EINVAL;
    for (int i = 0; i < am.n; ++i) {               // (the records must lie outside the zero-filled ranges, as x_meta)
        const char* m = (const char*)am.d[i].meta;
        if ((m >= (char*)z0 && m < (char*)z0 + bytes0) || (m >= (char*)z1 && m < (char*)z1 + bytes1)) return NEMO_EINVAL;
    }
    const int32_t rc = phase_fwd_args(&a, N, V, T, K, D, C, view_idx, frame_idx, raw_phase, shifts, scales, ldp, log_sigmas,
                                      codes, code_noise, kernel_id, X, ldx, phase_out, den_out, x_meta);
    if (rc) return rc;
    // (x_meta must not lie in [z0, z0 + bytes0) / [z1, z1 + bytes1): the zero-fill blocks of this launch run beside the phase blocks)
    if (x_meta && (((char*)x_meta >= (char*)z0 && (char*)x_meta < (char*)z0 + bytes0) || ((char*)x_meta >= (char*)z1 && (char*)x_meta < (char*)z1 + bytes1)))
        return NEMO_EINVAL;
    if (bytes0 < 0 || bytes1 < 0 || (bytes0 && !z0) || (bytes1 && !z1) || ((bytes0 | bytes1) & 3) ||
        (((uintptr_t)z0 | (uintptr_t)z1) & 15) || n_seg < 0 || n_seg > NEMO_ADAM_MAX_SEG)
        return NEMO_EINVAL;
    if (!segs_dev) n_seg = 0;
    const long n0 = bytes0 / 16, n1 = bytes1 / 16;
    int bz = nemo_cdiv((n0 > n1 ? n0 : n1), 256 * 4);
    if (bz < 1) bz = 1;
    if (bz > 1024) bz = 1024;
    const int nbp = (int)nemo_cdiv(N + 1, 4);
    StepBeginArgs b{(float4*)z0, n0, (float4*)z1, n1, (float*)z0 + 4 * n0, (int)((bytes0 & 15) / 4), (float*)z1 + 4 * n1,
                    (int)((bytes1 & 15) / 4), n_seg ? segs_dev : nullptr, (int)n_seg, beta1, beta2};
    hipLaunchKernelGGL(phase_embed_begin_kernel, dim3(nb_abs + nbp + bz), dim3(256), 0, (hipStream_t)stream, a, b, nbp, nb_abs, am);
    NEMO_LAUNCH_CHECK();
    return NEMO_OK;
}

// The whole phase / RBF / code backward in ONE launch (it sits at the end of the dX chain of every update step, and three
// dependent launches of 4 - 12 us each were 24 us of it).  Blocks [0, nAB): one per (view, slice of 64 phase-network nodes) --
// d shifts / d scales of that view's nodes over ALL samples of the view, in sample order (round 5: deterministic; see below).
// Blocks [nAB, ...): stage C (column sums of dX: d log_sigma, d code).
namespace {
constexpr int PH_LMAX = 1024;               // samples of a view per pass
constexpr int PH_DMAX = 128;                // RBF features whose constants are tabulated in LDS
constexpr int PH_CU = 10;                   // stage C: rows per pass of a column block (a 2401-row column at 256 threads: one pass)
// The backward coefficients of ONE sample, computed by the four lanes l = 0 .. 3 of a quad (all four return the result): d phase
// over the D RBF features -- a lane's <= 32 feature gradients are requested together: one memory latency --, then, with
// ph = num / den saved by the forward pass, (x, dy / K, (-dy - dden) / K, dden / K).
__device__ __forceinline__ float4 phase_sample_coef(int l, long s, long T, int K, int D, int kid, const int64_t* __restrict__ frame_idx,
                                                    const float* __restrict__ raw_phase, const float* __restrict__ log_sigmas,
                                                    const float* __restrict__ phase, const float* __restrict__ den_ws, float den_v,
                                                    const float* __restrict__ dX, long ldx, const float* t_ies, const float* t_ctr) {
    const float x = raw_phase ? raw_phase[s] : lin01(frame_idx[s], T);
    const float ph = phase[s];
    const float* g = dX + s * ldx;
    float dph = 0.f;
    if (D > 0 && D <= PH_DMAX) {
        float gv[PH_DMAX / 4];
#pragma unroll
        for (int u = 0; u < PH_DMAX / 4; ++u) gv[u] = (l + 4 * u) < D ? g[l + 4 * u] : 0.f;
        // (the kernel id is a launch constant: one switch per sample, not one per feature)
        auto feats = [&](auto kidc) {
#pragma unroll
            for (int u = 0; u < PH_DMAX / 4; ++u) {
                const int d = l + 4 * u;
                if (d < D) {
                    const float diff = ph - t_ctr[d], ies = t_ies[d];
                    dph += gv[u] * rbf_dphi(decltype(kidc)::value, (diff * diff) * ies) * 2.f * diff * ies;
                }
            }
        };
        switch (kid) {
            case 0: feats(std::integral_constant<int, 0>{}); break;
            case 1: feats(std::integral_constant<int, 1>{}); break;
            case 2: feats(std::integral_constant<int, 2>{}); break;
            case 3: feats(std::integral_constant<int, 3>{}); break;
            case 4: feats(std::integral_constant<int, 4>{}); break;
            case 5: feats(std::integral_constant<int, 5>{}); break;
            case 6: feats(std::integral_constant<int, 6>{}); break;
            case 7: feats(std::integral_constant<int, 7>{}); break;
            case 8: feats(std::integral_constant<int, 8>{}); break;
            case 9: feats(std::integral_constant<int, 9>{}); break;
            default: feats(std::integral_constant<int, 10>{}); break;
        }
    } else if (D > 0) {
        for (int d = l; d < D; d += 4) {
            const float diff = ph - lin01(d, D);
            const float ies = 1.f / expf(log_sigmas[d]);
            dph += g[d] * rbf_dphi(kid, (diff * diff) * ies) * 2.f * diff * ies;
        }
    } else if (l == 0) {
        dph = g[0];
    }
    dph += __shfl_xor(dph, 1, 64);
    dph += __shfl_xor(dph, 2, 64);
    const float den = den_ws ? den_ws[s] : den_v;
    const float dy = dph / den;                      // ph = num / den
    const float dden = -dy * ph;                     // -dph * num / den^2
    const float invK = 1.f / (float)K;
    return make_float4(x, dy * invK, (-dy - dden) * invK, dden * invK);
}
__device__ __forceinline__ void phase_bwd_fused_body(
    const int bid,
    long N, long V, long T, int K, int D, int C, const int64_t* __restrict__ view_idx,
    const int64_t* __restrict__ frame_idx, const float* __restrict__ raw_phase, const float* __restrict__ shifts,
    const float* __restrict__ scales, long ldp, const float* __restrict__ log_sigmas, int kid,
    const float* __restrict__ phase, const float* __restrict__ den_ws, const float* __restrict__ dX, long ldx,
    float* __restrict__ d_shifts, float* __restrict__ d_scales, float* __restrict__ d_log_sigmas,
    float* __restrict__ d_codes, int nAB, int sorted, int mode, float4* __restrict__ coef) {
    __shared__ float red[16];
    if (bid >= nAB) {
        // ---- stage C: one block per reduced column.  b < D: d log_sigma_d over all N + 1 rows; otherwise (c, v):
        // d code[v][c] over the samples of view v
        const int b = bid - nAB;
        float acc = 0.f;
        if (b < D) {
            if (!d_log_sigmas) return;
            const int d = b;
            const float es = expf(log_sigmas[d]), c = lin01(d, D);
            // PH_CU rows per pass, their loads UNCONDITIONAL (indices clamped into the matrix) and in flight together, the terms added in
            // row order as before: behind a predicate per row hipcc emits load -> wait -> branch, one round trip per row, and a 2401-row
            // column was ten dependent trips per thread -- the long pole of this launch at the tail of every step
            for (long s0 = threadIdx.x; s0 <= N; s0 += (long)blockDim.x * PH_CU) {
                float ph[PH_CU], g[PH_CU];
#pragma unroll
                for (int u = 0; u < PH_CU; ++u) {
                    const long sc = min(s0 + (long)blockDim.x * u, N);
                    ph[u] = phase[min(sc, N > 0 ? N - 1 : 0)];
                    g[u] = dX[sc * ldx + d];
                }
#pragma unroll
                for (int u = 0; u < PH_CU; ++u) {
                    const long su = s0 + (long)blockDim.x * u;
                    if (su > N) continue;
                    const float diff = (su < N ? ph[u] : 0.f) - c;
                    const float a = (diff * diff) / es;
                    acc -= g[u] * rbf_dphi(kid, a) * a;             // d a / d log_sigma = -a
                }
            }
            const float t = block_sum(acc, red);
            if (threadIdx.x == 0) d_log_sigmas[d] += t;
        } else {
            const int idx = b - D;
            const int c = idx % C;
            const long v = idx / C;
            const int off = D > 0 ? D : 1;
            for (long s0 = threadIdx.x; s0 < N; s0 += (long)blockDim.x * PH_CU) {
                long vi[PH_CU];
                float g[PH_CU];
#pragma unroll
                for (int u = 0; u < PH_CU; ++u) {
                    const long sc = min(s0 + (long)blockDim.x * u, N - 1);
                    vi[u] = view_idx[sc];
                    g[u] = dX[sc * ldx + off + c];
                }
#pragma unroll
                for (int u = 0; u < PH_CU; ++u)
                    if (s0 + (long)blockDim.x * u < N && vi[u] == v) acc += g[u];
            }
            const float t = block_sum(acc, red);
            if (threadIdx.x == 0) d_codes[v * C + c] += t;
        }
        return;
    }
    // ---- the phase network's gradient.  Until round 4: blocks of 32 samples, per-sample coefficients in LDS, one float atomic
    // per (run of samples of a view, node) -- the order of the additions, and the last bits of every gradient entry, changed
    // from run to run.  Now a block (v, ks) owns a slice of the nodes of view v outright: it finds the view's samples (a 64-ary
    // search when the caller vouches for a batch sorted by view, otherwise an ordered compaction of a scan over the batch),
    // takes their coefficients -- d phase over the D RBF features; with ph = num / den saved by the forward pass only
    // den = o - z + 1e-6 is needed, a per-view constant the forward kernel hands over in den_ws -- and lets thread (node kl,
    // sample lane sl) add samples sl, sl + NSL, ... in order; the sample lanes are combined in a fixed order.  No cross-block
    // sum is left.  mode 1 + mode 2 (two launches): the coefficients are computed once per sample by blocks of 64 samples and
    // read back here; mode 0 (no scratch or no den_ws): every (view, slice) block recomputes its view's coefficients.
    // (node-sum launch: 16 nodes x 16 sample lanes per block -- four times the blocks, a quarter of the serial loop)
    const int NK = mode == 2 ? 16 : 64, NSL = 256 / NK;
    const int KS = (K + NK - 1) / NK;
    const long v = bid / KS;
    const int ks = bid % KS, kl = threadIdx.x % NK, sl = threadIdx.x / NK, lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int k = ks * NK + kl;
    const bool kok = k < K;
    __shared__ int list[PH_LMAX];
    __shared__ float cx[PH_LMAX], cy[PH_LMAX], cz[PH_LMAX], co[PH_LMAX];
    __shared__ float t_ies[PH_DMAX], t_ctr[PH_DMAX];
    __shared__ int wcnt[4];
    __shared__ long rng[2];
    __shared__ float gred[2][256];
    if (mode != 2)
        for (int d = threadIdx.x; d < D && d < PH_DMAX; d += 256) { t_ies[d] = 1.f / expf(log_sigmas[d]); t_ctr[d] = lin01(d, D); }
    if (mode == 1) {
        // launch 1 of 2 (mode 2 follows on the same stream): this block's 64 samples -> their coefficients, once, in `coef`.
        // (Computed inside the per-view blocks below -- mode 0 -- a 300-sample view is five sequential passes per block and the
        //  8 x 300 step was 45 us longer.)
        __syncthreads();
        const long s = (long)bid * 64 + (threadIdx.x >> 2);
        const int l = threadIdx.x & 3;
        const bool live = s < N;
        const float4 c4 = phase_sample_coef(l, live ? s : 0, T, K, D, kid, frame_idx, raw_phase, log_sigmas, phase, den_ws, 1.f, dX, ldx,
                                            t_ies, t_ctr);
        if (live && l == 0) coef[s] = c4;
        return;
    }
    const float shr = kok ? shifts[v * ldp + k] : 0.f, scr = kok ? scales[v * ldp + k] : 0.f;
    const float shp = fmaxf(shr, 0.f), scp = fmaxf(scr, 0.f);
    const float s0v = sigmoidf_(scp * (0.f - shp)), s1v = sigmoidf_(scp * (1.f - shp));
    const float d0 = s0v * (1.f - s0v), d1 = s1v * (1.f - s1v);
    float den_v = 1.f;
    if (mode == 0 && !den_ws) {                              // (no forward hand-over: the view's two sigmoid sums)
        float z = 0.f, o = 0.f;
        for (int q = threadIdx.x; q < K; q += 256) {
            const float a_ = fmaxf(shifts[v * ldp + q], 0.f), b_ = fmaxf(scales[v * ldp + q], 0.f);
            z += sigmoidf_(b_ * (0.f - a_));
            o += sigmoidf_(b_ * (1.f - a_));
        }
        z = block_sum(z, red);
        __syncthreads();
        o = block_sum(o, red);
        if (threadIdx.x == 0) red[15] = o / (float)K - z / (float)K + 1e-6f;
        __syncthreads();
        den_v = red[15];
    }
    __syncthreads();
    float gsh = 0.f, gsc = 0.f;
    // `cnt` samples: base + i (a batch sorted by view) or list[i] (gathered)
    auto process = [&](int cnt, long base) {
        if (mode == 2) {                                     // coefficients from launch 1
            for (int i = threadIdx.x; i < cnt; i += 256) {
                const float4 c4 = coef[base >= 0 ? base + i : (long)list[i]];
                cx[i] = c4.x; cy[i] = c4.y; cz[i] = c4.z; co[i] = c4.w;
            }
        } else {                                             // four lanes per sample, 64 samples per pass
            for (int i0 = 0; i0 < cnt; i0 += 64) {
                const int i = i0 + (threadIdx.x >> 2), l = threadIdx.x & 3;
                const bool live = i < cnt;
                const long s = base >= 0 ? base + (live ? i : 0) : (long)list[live ? i : 0];
                const float4 c4 = phase_sample_coef(l, s, T, K, D, kid, frame_idx, raw_phase, log_sigmas, phase, den_ws, den_v, dX, ldx,
                                                    t_ies, t_ctr);
                if (l == 0 && live) { cx[i] = c4.x; cy[i] = c4.y; cz[i] = c4.z; co[i] = c4.w; }
            }
        }
        __syncthreads();
        if (kok)
            for (int i = sl; i < cnt; i += NSL) {            // relu'(0) = 0 as in torch: applied when the sums are written
                const float x = cx[i];
                const float sx = sigmoidf_(scp * (x - shp));
                const float wy = cy[i] * sx * (1.f - sx), wz = cz[i] * d0, wo = co[i] * d1;
                gsc += wy * (x - shp) + wz * (0.f - shp) + wo * (1.f - shp);
                gsh -= (wy + wz + wo) * scp;
            }
        __syncthreads();
    };
    if (sorted) {
        // the caller vouches for a batch sorted by view (every full batch): the view's range by a 64-ary search per bound
        if (threadIdx.x < 128) {
            const long want = v + (threadIdx.x >> 6);        // wave 0: first index with view >= v; wave 1: >= v + 1
            long lo = 0, hi = N;                             // answer in [lo, hi]
            while (hi - lo > 0) {
                const long step = (hi - lo + 63) / 64;
                const long pos = lo + (long)lane * step;     // lanes probe 64 positions; the predicate is monotone
                const bool ge = pos >= hi || view_idx[pos] >= want;
                const unsigned long long m = __ballot(ge);
                const int first = m ? __ffsll((long long)m) - 1 : 64;
                const long nlo = first == 0 ? lo : lo + (long)(first - 1) * step + 1;
                const long nhi = first == 64 ? hi : min(hi, lo + (long)first * step);
                if (first == 0) { hi = lo; } else { lo = nlo; hi = nhi; }
            }
            if (lane == 0) rng[threadIdx.x >> 6] = lo;
        }
        __syncthreads();
        const long lo = rng[0], hi = rng[1];
        for (long b0 = lo; b0 < hi; b0 += PH_LMAX) process((int)min((long)PH_LMAX, hi - b0), b0);
    } else {
        int n = 0;
        for (long base = 0; base < N; base += 256) {
            const long s = base + threadIdx.x;
            const bool hit = s < N && view_idx[s] == v;
            const unsigned long long m = __ballot(hit);
            if (lane == 0) wcnt[wv] = __popcll(m);
            __syncthreads();
            int off = n;
            for (int w = 0; w < wv; ++w) off += wcnt[w];
            if (hit) list[off + __popcll(m & ((1ull << lane) - 1ull))] = (int)s;
            n += wcnt[0] + wcnt[1] + wcnt[2] + wcnt[3];
            __syncthreads();
            if (n + 256 > PH_LMAX || base + 256 >= N) {
                if (n > 0) process(n, -1);
                n = 0;
            }
        }
    }
    gred[0][threadIdx.x] = gsh; gred[1][threadIdx.x] = gsc;
    __syncthreads();
    if (sl == 0 && kok) {
        float tsh = 0.f, tsc = 0.f;
        for (int q = 0; q < NSL; q += 4) {                   // the sample lanes in a fixed order
            tsh += (gred[0][q * NK + kl] + gred[0][(q + 1) * NK + kl]) + (gred[0][(q + 2) * NK + kl] + gred[0][(q + 3) * NK + kl]);
            tsc += (gred[1][q * NK + kl] + gred[1][(q + 1) * NK + kl]) + (gred[1][(q + 2) * NK + kl] + gred[1][(q + 3) * NK + kl]);
        }
        if (shr > 0.f && tsh != 0.f) d_shifts[v * ldp + k] += tsh;
        if (scr > 0.f && tsc != 0.f) d_scales[v * ldp + k] += tsc;
    }
}
struct PhaseBwdArgs {
    long N, V, T; int K, D, C; const int64_t* view_idx; const int64_t* frame_idx; const float* raw_phase;
    const float* shifts; const float* scales; long ldp; const float* log_sigmas; int kid; const float* phase;
    const float* den_ws; const float* dX; long ldx; float* d_shifts; float* d_scales; float* d_log_sigmas; float* d_codes;
    int nAB, sorted, mode; float4* coef;
};
#define PHASE_BWD_CALL(a, bid) phase_bwd_fused_body(bid, a.N, a.V, a.T, a.K, a.D, a.C, a.view_idx, a.frame_idx, a.raw_phase, \
    a.shifts, a.scales, a.ldp, a.log_sigmas, a.kid, a.phase, a.den_ws, a.dX, a.ldx, a.d_shifts, a.d_scales, a.d_log_sigmas, \
    a.d_codes, a.nAB, a.sorted, a.mode, a.coef)
__global__ __launch_bounds__(256) void phase_bwd_fused_kernel(PhaseBwdArgs a) { PHASE_BWD_CALL(a, (int)blockIdx.x); }

// The phase backward with the step's batched bias column sums (nemo_colsum_multi: out[n] += sum_m X[m][n] for up to
// NEMO_COLSUM_MAX matrices) in further blocks of the same grid: both only need the activation gradients the dX chain has
// produced, and as a launch of its own the column-sum pass sat behind the last parameter-gradient GEMM on the side stream, at
// the very end of the backward.
struct ColsumBatchP { nemo_colsum_desc d[NEMO_COLSUM_MAX]; int n; int gx, gy; long rows_per_block; NemoRed rr; };
__global__ __launch_bounds__(256) void phase_bwd_colsum_kernel(PhaseBwdArgs a, int n_phase, ColsumBatchP cb) {
    if ((int)blockIdx.x < n_phase) { PHASE_BWD_CALL(a, (int)blockIdx.x); return; }
    const int lin = (int)blockIdx.x - n_phase;
    const int bx = lin % cb.gx, by = (lin / cb.gx) % cb.gy, bz = lin / (cb.gx * cb.gy);
    const nemo_colsum_desc d = cb.d[bz];
    const long n = (long)bx * 64 + (threadIdx.x & 63);
    const long mbeg = (long)by * cb.rows_per_block;
    if ((long)bx * 64 >= d.N || d.M <= 0) return;                    // block-uniform (chunks beyond a shorter matrix deposit zeros)
    const long mend = min((long)d.M, mbeg + cb.rows_per_block);
    float sacc = 0.f;
    if (n < d.N) {
        // four independent accumulators: the loads of a wave's rows are in flight together (one dependent chain of 32 loads
        // per thread made this part a 30 us tail of the small-batch backward)
        float s1 = 0.f, s2 = 0.f, s3 = 0.f;
        long m = mbeg + (threadIdx.x >> 6);
        for (; m + 12 < mend; m += 16) {
            sacc += d.X[m * d.ldx + n]; s1 += d.X[(m + 4) * d.ldx + n];
            s2 += d.X[(m + 8) * d.ldx + n]; s3 += d.X[(m + 12) * d.ldx + n];
        }
        for (; m < mend; m += 4) sacc += d.X[m * d.ldx + n];
        sacc += s1 + s2 + s3;
    }
    __shared__ float redc[4][64];
    __shared__ int rflagc;
    redc[threadIdx.x >> 6][threadIdx.x & 63] = sacc;
    __syncthreads();
    const float t = threadIdx.x < 64 ? redc[0][threadIdx.x] + redc[1][threadIdx.x] + redc[2][threadIdx.x] + redc[3][threadIdx.x] : 0.f;
    const int strip = bz * cb.gx + bx;
    nemo_colsum_finish(t, n, d.N, d.out, cb.rr, strip, cb.gx * cb.n, by, cb.gy, strip, &rflagc);
}
}  // namespace

static int32_t phase_bwd_launch(int64_t N, int64_t V, int64_t T, int64_t K, int64_t D, int64_t C,
                                const int64_t* view_idx, const int64_t* frame_idx, const float* raw_phase,
                                const float* shifts, const float* scales, int64_t ldp, const float* log_sigmas,
                                int32_t kernel_id, const float* phase, const float* dX, int64_t ldx, float* ws,
                                float* d_shifts, float* d_scales, float* d_log_sigmas, float* d_codes, int32_t n_cs,
                                const nemo_colsum_desc* descs, int32_t sorted_by_view, void* stream) {
    if (N < 0 || V <= 0 || K <= 0 || D < 0 || C < 0 || !dX || !shifts || !scales || !phase) return NEMO_EINVAL;
    if ((d_shifts == nullptr) != (d_scales == nullptr)) return NEMO_EINVAL;
    if (n_cs < 0 || n_cs > NEMO_COLSUM_MAX || (n_cs && !descs)) return NEMO_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    // blocks [0, nAB): one per (view, 64-node slice) of the phase networks -- or, when the forward pass handed the views'
    // denominators over in ws and the library scratch has room, one per 64 samples computing their coefficients, the
    // (view, slice) blocks following as a second launch that only runs the node sums.
    long nNodes = (d_shifts && N > 0) ? V * ((K + 63) / 64) : 0;
    float4* coef = nullptr;
    if (nNodes && ws) coef = reinterpret_cast<float4*>(nemo_red_take((size_t)N * 4, 1).part);
    if (coef) nNodes = V * ((K + 15) / 16);                  // (the second launch's blocks take 16 nodes each)
    const long nAB = coef ? nemo_cdiv(N, 64) : nNodes;
    // blocks [nAB, nAB + D) reduce log_sigma columns; if d_log_sigmas is NULL they are still launched (as no-ops) so
    // that the block -> column map stays fixed
    const long nC = (d_log_sigmas || d_codes) ? D + (d_codes ? V * C : 0) : 0;
    PhaseBwdArgs a{(long)N, (long)V, (long)T, (int)K, (int)D, (int)C, view_idx, frame_idx, raw_phase, shifts, scales, (long)ldp,
                   log_sigmas, (int)kernel_id, phase, (const float*)ws, dX, (long)ldx, d_shifts, d_scales, d_log_sigmas, d_codes,
                   (int)nAB, sorted_by_view ? 1 : 0, coef ? 1 : 0, coef};
    ColsumBatchP cb;
    cb.n = 0; cb.gx = cb.gy = 0; cb.rows_per_block = 64; cb.rr = NemoRed{nullptr, nullptr};
    long maxM = 0, maxN = 0;
    for (int i = 0; i < n_cs; ++i) {
        if (descs[i].M < 0 || descs[i].N < 0 || !descs[i].X || !descs[i].out) return NEMO_EINVAL;
        cb.d[i] = descs[i];
        if (descs[i].M > maxM) maxM = descs[i].M;
        if (descs[i].N > maxN) maxN = descs[i].N;
    }
    long ncs = 0;
    if (n_cs && maxM > 0 && maxN > 0) {
        if (maxM > 4096) cb.rows_per_block = 256;            // (tall matrices: fewer row chunks for a strip's last arriver to add up)
        cb.n = n_cs; cb.gx = (int)nemo_cdiv(maxN, 64); cb.gy = (int)nemo_cdiv(maxM, cb.rows_per_block);
        ncs = (long)cb.gx * cb.gy * n_cs;
        cb.rr = nemo_red_take((size_t)ncs * 64, cb.gx * n_cs);
    }
    if (nAB + nC + ncs == 0) return NEMO_OK;
    if (ncs == 0)
        hipLaunchKernelGGL(phase_bwd_fused_kernel, dim3((unsigned)(nAB + nC)), dim3(256), 0, st, a);
    else
        hipLaunchKernelGGL(phase_bwd_colsum_kernel, dim3((unsigned)(nAB + nC + ncs)), dim3(256), 0, st, a, (int)(nAB + nC), cb);
    NEMO_LAUNCH_CHECK();
    if (coef) {
        a.nAB = (int)nNodes;
        a.mode = 2;
        hipLaunchKernelGGL(phase_bwd_fused_kernel, dim3((unsigned)nNodes), dim3(256), 0, st, a);
        NEMO_LAUNCH_CHECK();
    }
    return NEMO_OK;
}

extern "C" int32_t nemo_phase_embed_bwd(int64_t N, int64_t V, int64_t T, int64_t K, int64_t D, int64_t C,
                                        const int64_t* view_idx, const int64_t* frame_idx,
                                        const float* raw_phase, const float* shifts, const float* scales,
                                        int64_t ldp, const float* log_sigmas, int32_t kernel_id,
                                        const float* phase, const float* dX, int64_t ldx, float* ws,
                                        float* d_shifts, float* d_scales, float* d_log_sigmas,
                                        float* d_codes, int32_t sorted_by_view, void* stream) {
    return phase_bwd_launch(N, V, T, K, D, C, view_idx, frame_idx, raw_phase, shifts, scales, ldp, log_sigmas, kernel_id, phase,
                            dX, ldx, ws, d_shifts, d_scales, d_log_sigmas, d_codes, 0, nullptr, sorted_by_view, stream);
}

// nemo_phase_embed_bwd + nemo_colsum_multi(n_cs, descs) in ONE launch (further blocks of the same grid).
extern "C" int32_t nemo_phase_embed_bwd_colsum(int64_t N, int64_t V, int64_t T, int64_t K, int64_t D, int64_t C,
                                               const int64_t* view_idx, const int64_t* frame_idx, const float* raw_phase,
                                               const float* shifts, const float* scales, int64_t ldp,
                                               const float* log_sigmas, int32_t kernel_id, const float* phase,
                                               const float* dX, int64_t ldx, float* ws, float* d_shifts, float* d_scales,
                                               float* d_log_sigmas, float* d_codes, int32_t n_cs,
                                               const nemo_colsum_desc* descs, int32_t sorted_by_view, void* stream) {
    return phase_bwd_launch(N, V, T, K, D, C, view_idx, frame_idx, raw_phase, shifts, scales, ldp, log_sigmas, kernel_id, phase,
                            dX, ldx, ws, d_shifts, d_scales, d_log_sigmas, d_codes, n_cs, descs, sorted_by_view, stream);
}

extern "C" int32_t nemo_rot6d_fwd(int64_t rows, int64_t J, const float* rot6d, int64_t ld6, int32_t zero_nan,
                                  float* R, float* aa, void* stream) {
    if (rows < 0 || J <= 0 || !rot6d || ld6 < J * 6) return NEMO_EINVAL;
    if (rows == 0) return NEMO_OK;
    hipLaunchKernelGGL(rot6d_fwd_kernel, GRID1D(rows * J), (long)(rows * J), (int)J, rot6d, (long)ld6,
                       (int)zero_nan, R, aa);
    NEMO_LAUNCH_CHECK();
    return NEMO_OK;
}

extern "C" int32_t nemo_rot6d_bwd(int64_t rows, int64_t J, const float* rot6d, int64_t ld6, int32_t zero_nan,
                                  const float* dR, const float* daa, float* d_rot6d, int64_t ldd,
                                  void* stream) {
    if (rows < 0 || J <= 0 || !rot6d || !d_rot6d || ld6 < J * 6 || ldd < J * 6) return NEMO_EINVAL;
    if (rows == 0) return NEMO_OK;
    hipLaunchKernelGGL(rot6d_bwd_kernel, GRID1D(rows * J), (long)(rows * J), (int)J, rot6d, (long)ld6,
                       (int)zero_nan, dR, daa, d_rot6d, (long)ldd);
    NEMO_LAUNCH_CHECK();
    return NEMO_OK;
}


extern "C" int32_t nemo_pose_bwd_fused(int64_t N, const float* rot6d, int64_t ld6, int32_t zero_nan,
                                       const float* dR, const float* daa, float* d_rot6d, int64_t ldd,
                                       const float* aa, const float* dR2, float v2v_scale,
                                       float* dTR, int64_t ldt, int32_t zero_row, float* head_meta, void* stream) {
    if (N < 0 || !rot6d || !d_rot6d || ld6 < 144 || ldd < 144 || (dR2 && !aa) || (dTR && ldt < 3))
        return NEMO_EINVAL;
    if (N == 0) return NEMO_OK;
    const unsigned blocks = (unsigned)nemo_cdiv(N * 24, 256) + 1;          // + the reduction block
    hipLaunchKernelGGL(pose_bwd_fused_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (long)N, rot6d,
                       (long)ld6, (int)zero_nan, dR, daa, d_rot6d, (long)ldd, aa, dR2, v2v_scale, dTR, (long)ldt,
                       (int)zero_row, head_meta);
    NEMO_LAUNCH_CHECK();
    return NEMO_OK;
}

extern "C" int32_t nemo_rotmat_to_aa(int64_t M, const float* R, int32_t zero_nan, float* aa, void* stream) {
    if (M < 0 || !R || !aa) return NEMO_EINVAL;
    if (M == 0) return NEMO_OK;
    hipLaunchKernelGGL(rotmat_to_aa_kernel, GRID1D(M), (long)M, R, (int)zero_nan, aa);
    NEMO_LAUNCH_CHECK();
    return NEMO_OK;
}

extern "C" int32_t nemo_rodrigues_fwd(int64_t M, const float* theta, int32_t form, float* R, void* stream) {
    if (M < 0 || !theta || !R || form < 0 || form > 1) return NEMO_EINVAL;
    if (M == 0) return NEMO_OK;
    hipLaunchKernelGGL(rodrigues_fwd_kernel, GRID1D(M), (long)M, theta, (int)form, R);
    NEMO_LAUNCH_CHECK();
    return NEMO_OK;
}

extern "C" int32_t nemo_rodrigues_bwd(int64_t M, const float* theta, const float* dR, float* dtheta,
                                      void* stream) {
    if (M < 0 || !theta || !dR || !dtheta) return NEMO_EINVAL;
    if (M == 0) return NEMO_OK;
    hipLaunchKernelGGL(rodrigues_bwd_kernel, GRID1D(M), (long)M, theta, dR, dtheta);
    NEMO_LAUNCH_CHECK();
    return NEMO_OK;
}

extern "C" int32_t nemo_v2v_prep_fwd(int64_t N, const float* R, const float* aa, const float* aa_dec,
                                     float* R2, const int64_t* n_valid, void* stream) {
    if (N < 0 || !R || !aa || !aa_dec || !R2) return NEMO_EINVAL;
    if (N == 0) return NEMO_OK;
    hipLaunchKernelGGL(v2v_prep_fwd_kernel, GRID1D(2 * N * 24), (long)N, R, aa, aa_dec, R2, n_valid);
    NEMO_LAUNCH_CHECK();
    return NEMO_OK;
}

extern "C" int32_t nemo_v2v_prep_fwd_dec(int64_t N, const float* R, const float* aa, const float* dec6d, int64_t lddec,
                                         float* aa_dec_out, float* R2, const int64_t* n_valid, void* stream) {
    if (N < 0 || !R || !aa || !dec6d || lddec < 126 || !aa_dec_out || !R2) return NEMO_EINVAL;
    if (N == 0) return NEMO_OK;
    hipLaunchKernelGGL(v2v_prep_fwd_kernel, GRID1D(2 * N * 24), (long)N, R, aa, (const float*)nullptr, R2, n_valid, dec6d,
                       (long)lddec, aa_dec_out);
    NEMO_LAUNCH_CHECK();
    return NEMO_OK;
}

extern "C" int32_t nemo_v2v_prep_bwd(int64_t N, const float* aa, const float* dR2, float scale, float* d_aa,
                                     float* dR, void* stream) {
    if (N < 0 || !aa || !dR2 || !d_aa || !dR) return NEMO_EINVAL;
    if (N == 0) return NEMO_OK;
    hipLaunchKernelGGL(v2v_prep_bwd_kernel, GRID1D(N * 24), (long)N, aa, dR2, scale, d_aa, dR);
    NEMO_LAUNCH_CHECK();
    return NEMO_OK;
}

// Instance-code regulariser of NemoV3 / V4 (nemo/neural_motion_model.py:3864-3867): scalar_out += mean(x^2) and, when
// grad != NULL, grad += gscale * x  (gscale = 2 * weight / numel, times the shard's share of the views).  One block: the
// code table is V x C floats.  The sum is taken in thread order (deterministic).
namespace {
__global__ __launch_bounds__(256) void sqmean_kernel(long n, const float* __restrict__ x, float* __restrict__ scalar_out,
                                                     float* __restrict__ grad, float gscale) {
    __shared__ float red[16];
    float acc = 0.f;
    for (long i = threadIdx.x; i < n; i += 256) {
        const float v = x[i];
        acc += v * v;
        if (grad) grad[i] += gscale * v;
    }
    const float t = block_sum(acc, red);
    if (threadIdx.x == 0) *scalar_out += t / (float)n;
}
}  // namespace

extern "C" int32_t nemo_sqmean_fwd_bwd(int64_t n, const float* x, float* scalar_out, float* grad, float gscale,
                                       void* stream) {
    if (n <= 0 || !x || !scalar_out) return NEMO_EINVAL;
    hipLaunchKernelGGL(sqmean_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, (long)n, x, scalar_out, grad, gscale);
    NEMO_LAUNCH_CHECK();
    return NEMO_OK;
}

extern "C" int32_t nemo_scale_neg_rowsum(int64_t N, int64_t cols, const float* X, int64_t ldx, float* out_row,
                                         void* stream) {
    if (N < 0 || cols <= 0 || !X || !out_row) return NEMO_EINVAL;
    if (N == 0) return NEMO_OK;
    hipLaunchKernelGGL(neg_rowsum_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, (long)N, (int)cols,
                       X, (long)ldx, out_row);
    NEMO_LAUNCH_CHECK();
    return NEMO_OK;
}
