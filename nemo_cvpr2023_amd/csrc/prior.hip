// Pose priors and the fused optimiser: VPoser KL term, GMM max-mixture prior, robust 3-D pose loss,
// multi-segment Adam.  Each loss kernel produces its (upstream-free) gradient in the same pass.
#include <cstdlib>
#include <atomic>
#include "common.h"
#include "../../include/nemo_hip.h"

namespace {

// KL(N(mu, s) || N(0,1)) = 0.5 (s^2 + mu^2 - 1 - log s^2),  s = softplus(lv)
// (torch.distributions.kl._kl_normal_normal; vposer_model.py:55; nemo/neural_motion_model.py:2795-2802)
__global__ __launch_bounds__(256) void kl_kernel(long N, int L, const float* __restrict__ mulv, long ld,
                                                 float* __restrict__ out, float* __restrict__ d, long ldd,
                                                 const int64_t* __restrict__ n_valid, NemoRed rr) {
    __shared__ float red[16];
    __shared__ int rflag;
    float acc = 0.f;
    const long total = N * L;
    const float invN = 1.f / (float)N;
    const long nv = n_valid ? min((long)*n_valid, N) : N;      // rows >= nv are padding: no loss, zero gradient
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long s = i / L;
        const int k = (int)(i % L);
        if (s >= nv) {
            if (d) { d[s * ldd + k] = 0.f; d[s * ldd + L + k] = 0.f; }
            continue;
        }
        const float mu = mulv[s * ld + k], lv = mulv[s * ld + L + k];
        // F.softplus (beta=1, threshold=20)
        const float sp = lv > 20.f ? lv : log1pf(expf(lv));
        const float var = sp * sp;
        acc += 0.5f * (var + mu * mu - 1.f - logf(var));
        if (d) {
            const float dsp = sp - 1.f / sp;                         // d/ds 0.5(s^2 - log s^2)
            const float sig = lv > 20.f ? 1.f : 1.f / (1.f + expf(-lv));
            d[s * ldd + k] = mu * invN;
            d[s * ldd + L + k] = dsp * sig * invN;
        }
    }
    const float t = block_sum(acc, red);
    nemo_red_scalar(t * invN, out, rr, (int)blockIdx.x, (int)gridDim.x, red, &rflag);     // (deterministic: common.h)
}

// MaxMixturePrior.  Grid = (sample blocks of 128, mixture components): a block owns ONE component,
// stages its 69x69 precision matrix in LDS (rows padded to 72 floats so the broadcast reads are
// ds_read_b128) and the poses of its 128 samples transposed ([dim][128], conflict-free).  Every lane
// then runs the 69x69 quadratic form with LDS-broadcast operands.  Pass 1 writes the per-component
// log-likelihoods, pass 2 picks the arg-min and back-propagates only through the selected component.
#define GMM_TS 128
template <int DIM, bool SYM>
__device__ __forceinline__ void gmm_stage(const float* __restrict__ P, float (*Ps)[72]) {
    for (int idx = threadIdx.x; idx < DIM * DIM; idx += GMM_TS) {
        const int i = idx / DIM, j = idx % DIM;
        Ps[i][j] = SYM ? 0.5f * (P[i * DIM + j] + P[j * DIM + i]) : P[i * DIM + j];
    }
}

template <int DIM>
__global__ __launch_bounds__(GMM_TS) void gmm_ll_kernel(long N, int M, const float* __restrict__ x, long ldx,
                                                        const float* __restrict__ means,
                                                        const float* __restrict__ prec,
                                                        const float* __restrict__ log_nllw,
                                                        float* __restrict__ ll) {
    __shared__ __attribute__((aligned(16))) float Ps[DIM][72];
    __shared__ float xs[DIM][GMM_TS];
    const int tid = threadIdx.x, m = blockIdx.y;
    const long s = (long)blockIdx.x * GMM_TS + tid;
    const bool live = s < N;
    gmm_stage<DIM, false>(prec + (long)m * DIM * DIM, Ps);
    const float* mu = means + m * DIM;
    for (int i = 0; i < DIM; ++i) xs[i][tid] = (live ? x[s * ldx + i] : 0.f) - mu[i];
    __syncthreads();
    float d[DIM];
#pragma unroll
    for (int j = 0; j < DIM; ++j) d[j] = xs[j][tid];
    float qf = 0.f;
    for (int i = 0; i < DIM; ++i) {
        float y = 0.f;
#pragma unroll
        for (int j = 0; j < DIM; ++j) y += Ps[i][j] * d[j];       // einsum('mij,bmj->bmi'), prior.py:184
        qf += y * xs[i][tid];
    }
    if (live) ll[s * M + m] = 0.5f * qf - log_nllw[m];
}

template <int DIM>
__global__ __launch_bounds__(GMM_TS) void gmm_grad_kernel(long N, int M, const float* __restrict__ x, long ldx,
                                                          const float* __restrict__ means,
                                                          const float* __restrict__ prec,
                                                          const float* __restrict__ ll,
                                                          float* __restrict__ out,
                                                          float* __restrict__ per_sample, float coef,
                                                          float* __restrict__ dx, long lddx,
                                                          const int64_t* __restrict__ n_valid, NemoRed rr) {
    __shared__ __attribute__((aligned(16))) float Ps[DIM][72];
    __shared__ float xs[DIM][GMM_TS];
    __shared__ float red[16];
    __shared__ int any_sel;
    const int tid = threadIdx.x, m = blockIdx.y;
    const long s = (long)blockIdx.x * GMM_TS + tid;
    const bool live = s < (n_valid ? min((long)*n_valid, N) : N);     // (padding rows: no loss, no gradient)
    float best = 0.f;
    int best_m = -1;
    if (live) {
        best = ll[s * M]; best_m = 0;
        for (int k = 1; k < M; ++k) {                         // torch.min: first minimum wins
            const float v = ll[s * M + k];
            if (v < best) { best = v; best_m = k; }
        }
    }
    if (tid == 0) any_sel = 0;
    __syncthreads();
    if (m == 0) {                                             // loss value: one block row does it
        if (per_sample && live) per_sample[s] = best;
        const float tot = block_sum(live ? best : 0.f, red);
        __shared__ int rflag;
        nemo_red_scalar(tot / (float)N, out, rr, (int)blockIdx.x, (int)gridDim.x, red, &rflag);
    }
    const bool mine = live && best_m == m;
    if (mine) any_sel = 1;
    __syncthreads();
    if (!dx || !any_sel) return;                              // block-uniform
    gmm_stage<DIM, true>(prec + (long)m * DIM * DIM, Ps);     // 0.5 (P + P^T): adjoint of d^T P d / 2
    const float* mu = means + m * DIM;
    for (int i = 0; i < DIM; ++i) xs[i][tid] = (live ? x[s * ldx + i] : 0.f) - mu[i];
    __syncthreads();
    float d[DIM];
#pragma unroll
    for (int j = 0; j < DIM; ++j) d[j] = xs[j][tid];
    for (int i = 0; i < DIM; ++i) {
        float y = 0.f;
#pragma unroll
        for (int j = 0; j < DIM; ++j) y += Ps[i][j] * d[j];
        if (mine) dx[s * lddx + i] += coef * y;
    }
}


// MaxMixturePrior on the matrix cores (M <= 8 components).  Block = 16 samples x all components,
// wave w owns components w and w + 4.  y = P^T d (the quadratic form d^T y is that of P; y is its exact
// gradient for a symmetric P -- precisions are inverses of covariance matrices, and the engine
// symmetrises them on the host; reading P row-wise only keeps every operand load coalesced, the
// transposed read cost 25 us) as v_mfma_f32_16x16x4_f32 tiles: rows = 16 of the 69 outputs
// (5 tiles), columns = the 16 samples, K = 69 -> 18 steps; operands straight from L2 (the 8 precision
// matrices are 152 KB).  In the accumulator layout a lane holds y_i of ONE sample for 20 i's, so
// d^T y is 20 lane-local FMAs + two cross-lane adds, the arg-min over components goes through 512 B of
// LDS, and the gradient rows are written by the wave that owns the winning component -- one launch,
// no per-component scratch, ~8 us instead of two launches of ~40 us.
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void gmm_mfma_kernel(long N, int M, const float* __restrict__ x, long ldx,
                                                       const float* __restrict__ means,
                                                       const float* __restrict__ prec,
                                                       const float* __restrict__ log_nllw,
                                                       float* __restrict__ out, float* __restrict__ per_sample,
                                                       float coef, float* __restrict__ dx, long lddx,
                                                       const int64_t* __restrict__ n_valid, NemoRed rr) {
    constexpr int DIM = 69;
    __shared__ float llw[8][16];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, g = lane >> 4;
    const long s = (long)blockIdx.x * 16 + l15;
    const bool live = s < (n_valid ? min((long)*n_valid, N) : N);     // (padding rows: no loss, no gradient)
    const float* xr = x + (live ? s : 0) * ldx;
    // B-operand source: d[k][n] for k = 4 kk + g (zero beyond DIM); also d_i in the accumulator layout
    float xk[18];
#pragma unroll
    for (int kk = 0; kk < 18; ++kk) {
        const int k = 4 * kk + g;
        xk[kk] = k < DIM ? xr[k < DIM ? k : 0] : 0.f;
    }
    float xi[5][4];
#pragma unroll
    for (int ti = 0; ti < 5; ++ti)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int i = 16 * ti + 4 * g + r;
            xi[ti][r] = i < DIM ? xr[i < DIM ? i : 0] : 0.f;
        }
    f32x4 acc[2][5];
#pragma unroll
    for (int c = 0; c < 2; ++c) {
        const int m = wid + 4 * c;
        if (m >= M) {                                             // wave-uniform
            if (lane < 16) llw[(wid + 4 * c) & 7][lane] = 3.0e38f;
            continue;
        }
        const float* mu = means + m * DIM;
        const float* P = prec + (long)m * DIM * DIM;
#pragma unroll
        for (int ti = 0; ti < 5; ++ti)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[c][ti][r] = 0.f;
        // all 90 A-operands of the component first (180 independent L2 loads in flight: one latency), then
        // the 90 MFMAs back to back
        float a[18][5];
#pragma unroll
        for (int kk = 0; kk < 18; ++kk) {
            const int k = 4 * kk + g;
            const bool kok = k < DIM;
            const int kc = kok ? k : 0;
#pragma unroll
            for (int ti = 0; ti < 5; ++ti) {
                const int i = 16 * ti + l15;
                const int ic = i < DIM ? i : 0;
                const float v = P[kc * DIM + ic];       // row k, 16 consecutive columns per lane group: coalesced
                a[kk][ti] = (kok && i < DIM) ? v : 0.f;
            }
        }
        __builtin_amdgcn_sched_barrier(0);       // (hipcc otherwise re-sinks every load to its MFMA)
#pragma unroll
        for (int kk = 0; kk < 18; ++kk) {
            const int k = 4 * kk + g;
            const bool kok = k < DIM;
            const float b = kok ? xk[kk] - mu[kok ? k : 0] : 0.f;
#pragma unroll
            for (int ti = 0; ti < 5; ++ti)
                acc[c][ti] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[kk][ti], b, acc[c][ti], 0, 0, 0);
        }
        float q = 0.f;
#pragma unroll
        for (int ti = 0; ti < 5; ++ti)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int i = 16 * ti + 4 * g + r;
                const float di = i < DIM ? xi[ti][r] - mu[i < DIM ? i : 0] : 0.f;
                q += acc[c][ti][r] * di;
            }
        q += __shfl_xor(q, 16);
        q += __shfl_xor(q, 32);
        if (lane < 16) llw[m][lane] = 0.5f * q - log_nllw[m];
    }
    __syncthreads();
    float best = llw[0][l15];
    int best_m = 0;
    for (int k = 1; k < M; ++k) {                                 // torch.min: first minimum wins
        const float v = llw[k][l15];
        if (v < best) { best = v; best_m = k; }
    }
    {
        float t = 0.f;
        if (wid == 0) {
            if (per_sample && live && lane < 16) per_sample[s] = best;
            t = wave_sum((live && lane < 16) ? best : 0.f);
        }
        __shared__ float rred[16];
        __shared__ int rflag;
        nemo_red_scalar(t / (float)N, out, rr, (int)blockIdx.x, (int)gridDim.x, rred, &rflag);     // (thread 0 = wave 0, lane 0)
    }
    if (!dx || !live) return;
#pragma unroll
    for (int c = 0; c < 2; ++c) {
        if (best_m != wid + 4 * c) continue;
        float* row = dx + s * lddx;
#pragma unroll
        for (int ti = 0; ti < 5; ++ti)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int i = 16 * ti + 4 * g + r;
                if (i < DIM) row[i] += coef * acc[c][ti][r];
            }
    }
}

// mean over N*dim of (mask>0.5) * rho^2 r^2/(r^2+rho^2), r = x - target[view, frame]
__global__ __launch_bounds__(256) void pose3d_kernel(long N, int dim, const float* __restrict__ x, long ldx,
                                                     const float* __restrict__ target,
                                                     const float* __restrict__ mask,
                                                     const int64_t* __restrict__ view_idx,
                                                     const int64_t* __restrict__ frame_idx, long T,
                                                     float* __restrict__ out, float scale,
                                                     float* __restrict__ dx, long lddx,
                                                     const int64_t* __restrict__ n_valid, NemoRed rr) {
    __shared__ float red[16];
    __shared__ int rflag;
    const long total = N * dim;
    const float inv = 1.f / (float)total;
    const float rho2 = 10000.f;
    const long nv = n_valid ? min((long)*n_valid, N) : N;
    float acc = 0.f;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long s = i / dim;
        const int k = (int)(i % dim);
        if (s >= nv) continue;                                  // padding row
        const long vt = view_idx[s] * T + frame_idx[s];
        const float m = mask[vt] > 0.5f ? 1.f : 0.f;
        const float r = x[s * ldx + k] - target[vt * dim + k];
        const float r2 = r * r;
        acc += m * (rho2 * (r2 / (r2 + rho2)));
        if (dx) {
            const float den = r2 + rho2;
            dx[s * lddx + k] += scale * inv * m * 2.f * r * rho2 * rho2 / (den * den);
        }
    }
    const float t = block_sum(acc, red);
    nemo_red_scalar(t * inv, out, rr, (int)blockIdx.x, (int)gridDim.x, red, &rflag);
}

struct AdamSegs {
    nemo_adam_seg s[NEMO_ADAM_MAX_SEG];
    int n;
};

// torch.optim.Adam / AdamW single-tensor update rule (torch/optim/adam.py _single_tensor_adam):
//   grad += wd * p (Adam)  |  p *= 1 - lr*wd (AdamW)
//   m = lerp(m, g, 1-b1);  v = b2 v + (1-b2) g^2
//   p -= (lr / bc1) * m / (sqrt(v)/sqrt(bc2) + eps)
// SEGS_DEV: the segment table (learning rates, bias corrections: they change every step) is read from
// device memory, so a captured HIP graph of the step can be replayed with fresh values.
template <bool SEGS_DEV>
__global__ __launch_bounds__(256) void adam_kernel(AdamSegs segs, const nemo_adam_seg* __restrict__ segs_dev,
                                                   float* __restrict__ p, const float* __restrict__ g,
                                                   float* __restrict__ m, float* __restrict__ v, float b1,
                                                   float b2, float eps, const float* __restrict__ skip_if_nonzero) {
    if (skip_if_nonzero && *skip_if_nonzero != 0.f) return;        // (nemo_adam_step_dev_if: NaN gradients were counted)
    const nemo_adam_seg sg = SEGS_DEV ? segs_dev[blockIdx.y] : segs.s[blockIdx.y];
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < sg.numel; i += (long)gridDim.x * blockDim.x) {
        const long k = sg.offset + i;
        float grad = g[k], par = p[k];
        if (sg.weight_decay != 0.f) {
            if (sg.adamw) par *= 1.f - sg.lr * sg.weight_decay;
            else grad += sg.weight_decay * par;
        }
        const float mm = m[k] + (grad - m[k]) * (1.f - b1);
        const float vv = b2 * v[k] + (1.f - b2) * (grad * grad);
        m[k] = mm; v[k] = vv;
        const float denom = sqrtf(vv) / sg.bias_corr2_sqrt + eps;
        p[k] = par - sg.step_size * (mm / denom);
    }
}

// First launch of a step: the two zero-fills of the step (gradient buffer, per-workspace accumulator arena) and the
// per-update bookkeeping of the device-resident Adam table in ONE launch.
__global__ __launch_bounds__(256) void step_begin_kernel(float4* __restrict__ z0, long n0, float4* __restrict__ z1,
                                                         long n1, float* __restrict__ t0, int r0,
                                                         float* __restrict__ t1, int r1, nemo_adam_seg* segs, int n_seg,
                                                         double b1, double b2) {
    const long stride = (long)gridDim.x * blockDim.x;
    const long i0 = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
    for (long i = i0; i < n0; i += stride) z0[i] = z;
    for (long i = i0; i < n1; i += stride) z1[i] = z;
    if (i0 < r0) t0[i0] = 0.f;                                  // (< 4 trailing floats each)
    if (i0 < r1) t1[i0] = 0.f;
    if (segs && blockIdx.x == 0 && threadIdx.x < n_seg) {
        nemo_adam_seg sg = segs[threadIdx.x];
        sg.step += 1;
        sg.step_size = (float)((double)sg.lr / (1.0 - pow(b1, (double)sg.step)));
        sg.bias_corr2_sqrt = (float)sqrt(1.0 - pow(b2, (double)sg.step));
        segs[threadIdx.x] = sg;
    }
}

// Loss read-back without a host-side stream synchronisation: the values become final in the middle of the
// step (after the mesh kernel; the MLP backward and Adam follow), so one wave copies them to pinned,
// device-mapped host memory and raises a flag the host polls -- the host returns the losses and prepares
// the next launch while the rest of the step is still running.
__global__ void publish_kernel(const float* __restrict__ src, int n, float* __restrict__ host_dst,
                               int* __restrict__ host_flag) {
    const int t = threadIdx.x;
    if (t < n) __hip_atomic_store(host_dst + t, src[t], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __threadfence_system();
    if (t == 0) __hip_atomic_store(host_flag, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

}  // namespace

// ---- the CALLER-OWNED scratch of the ordered reductions (common.h; ABI 17) --------------------------------------------------
// Until ABI 16 this was one library-owned, process-global buffer behind unsynchronised globals (allocated on whichever device
// was current at the first nemo_ctx_create).  Now the caller owns the memory (nemo_reduce_ws_bytes / nemo_reduce_ws_bind) and
// the only state the library keeps is the CALLING THREAD's binding of such an arena and its bump cursor: engines on different
// devices / streams bind different arenas, host threads do not share a cursor.
namespace {
constexpr size_t RED_TICKETS = 1u << 16;          // ints at the head of the arena
struct RedArena { char* base = nullptr; size_t part_bytes = 0, off = 0, tk = 0; };
thread_local RedArena t_red;
std::atomic<long> g_red_fallbacks{0};
}  // namespace

NemoRed nemo_red_take(size_t part_floats, int n_tickets) {
    static const bool off = getenv("NEMO_ORDERED_REDUCE") != nullptr && atoi(getenv("NEMO_ORDERED_REDUCE")) == 0;
    if (off || n_tickets < 1) return NemoRed{nullptr, nullptr};
    RedArena& a = t_red;
    const size_t bytes = (part_floats * 4 + 255) / 256 * 256;
    if (!a.base || a.off + bytes > a.part_bytes || a.tk + (size_t)n_tickets > RED_TICKETS) {
        g_red_fallbacks.fetch_add(1, std::memory_order_relaxed);          // (no arena bound / exhausted: float atomics, counted)
        return NemoRed{nullptr, nullptr};
    }
    NemoRed r{reinterpret_cast<float*>(a.base + RED_TICKETS * 4 + a.off), reinterpret_cast<int*>(a.base) + a.tk};
    a.off += bytes;
    a.tk += (size_t)n_tickets;
    return r;
}

extern "C" int64_t nemo_reduce_ws_bytes(int64_t n_samples, int64_t n_views) {
    if (n_samples < 0 || n_views < 0) return -1;
    // tickets + deposits of one pass: key-point partials (3 launches), phase runs, column sums, scalars; generous and bounded
    const long per_sample = 48 * 4 * 3 + 16 + 4 * 64;                  // bytes
    long b = (long)RED_TICKETS * 4 + (8L << 20) + n_samples * per_sample + n_views * 4096;
    if (b > (long)RED_TICKETS * 4 + (160L << 20)) b = (long)RED_TICKETS * 4 + (160L << 20);
    return (b + 255) / 256 * 256;
}

extern "C" int32_t nemo_reduce_ws_bind(void* ws, int64_t bytes) {
    if (ws == nullptr) { t_red = RedArena{}; return NEMO_OK; }
    if ((((uintptr_t)ws) & 255) || bytes < (int64_t)(RED_TICKETS * 4 + 4096)) return NEMO_EINVAL;
    RedArena& a = t_red;
    a.base = (char*)ws;
    a.part_bytes = (size_t)bytes - RED_TICKETS * 4;
    a.off = 0;
    a.tk = 0;
    return NEMO_OK;
}

extern "C" int32_t nemo_reduce_scratch_reset(void) {
    t_red.off = 0;
    t_red.tk = 0;
    return NEMO_OK;
}

extern "C" int64_t nemo_reduce_fallbacks(void) { return g_red_fallbacks.load(std::memory_order_relaxed); }

extern "C" int32_t nemo_step_begin(void* z0, int64_t bytes0, void* z1, int64_t bytes1, nemo_adam_seg* segs_dev,
                                   int32_t n_seg, double beta1, double beta2, void* stream) {
    if (bytes0 < 0 || bytes1 < 0 || (bytes0 && !z0) || (bytes1 && !z1) || ((bytes0 | bytes1) & 3) ||
        (((uintptr_t)z0 | (uintptr_t)z1) & 15) || n_seg < 0 || n_seg > NEMO_ADAM_MAX_SEG)
        return NEMO_EINVAL;
    if (!segs_dev) n_seg = 0;
    if (bytes0 == 0 && bytes1 == 0 && n_seg == 0) return NEMO_OK;
    const long n0 = bytes0 / 16, n1 = bytes1 / 16;
    int blocks = nemo_cdiv((n0 > n1 ? n0 : n1), 256 * 4);
    if (blocks < 1) blocks = 1;
    if (blocks > 1024) blocks = 1024;
    hipLaunchKernelGGL(step_begin_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (float4*)z0, n0, (float4*)z1, n1,
                       (float*)z0 + 4 * n0, (int)((bytes0 & 15) / 4), (float*)z1 + 4 * n1, (int)((bytes1 & 15) / 4),
                       n_seg ? segs_dev : nullptr, (int)n_seg, beta1, beta2);
    NEMO_LAUNCH_CHECK();
    return NEMO_OK;
}

extern "C" int32_t nemo_publish_scalars(const float* src, int32_t n, float* host_dst, int32_t* host_flag,
                                        void* stream) {
    if (!src || !host_dst || !host_flag || n < 1 || n > 64) return NEMO_EINVAL;
    hipLaunchKernelGGL(publish_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, src, (int)n, host_dst,
                       (int*)host_flag);
    NEMO_LAUNCH_CHECK();
    return NEMO_OK;
}

extern "C" int32_t nemo_kl_fwd_bwd(int64_t N, int64_t L, const float* mulv, int64_t ld, float* scalar_out,
                                   float* d_mulv, int64_t ldd, const int64_t* n_valid, void* stream) {
    if (N <= 0 || L <= 0 || !mulv || !scalar_out || ld < 2 * L || (d_mulv && ldd < 2 * L)) return NEMO_EINVAL;
    int blocks = nemo_cdiv(N * L, 256);
    if (blocks > 1024) blocks = 1024;
    hipLaunchKernelGGL(kl_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (long)N, (int)L, mulv,
                       (long)ld, scalar_out, d_mulv, (long)ldd, n_valid, nemo_red_take(blocks, 1));
    NEMO_LAUNCH_CHECK();
    return NEMO_OK;
}

extern "C" int32_t nemo_gmm_fwd_bwd(int64_t N, int64_t M, int64_t dim, const float* x, int64_t ldx,
                                    const float* means, const float* precisions, const float* log_nllw,
                                    float* ws, float* scalar_out, float* per_sample, float scale, float* d_x,
                                    int64_t lddx, const int64_t* n_valid, void* stream) {
    if (N <= 0 || M <= 0 || !x || !means || !precisions || !log_nllw || !scalar_out || !ws || ldx < dim)
        return NEMO_EINVAL;
    if (dim != 69) return NEMO_EINVAL;   // SMPL body pose (23 joints x 3), prior.py:150
    if (d_x && lddx < dim) return NEMO_EINVAL;
    if (M <= 8) {
        hipLaunchKernelGGL(gmm_mfma_kernel, dim3(nemo_cdiv(N, 16)), dim3(256), 0, (hipStream_t)stream, (long)N,
                           (int)M, x, (long)ldx, means, precisions, log_nllw, scalar_out, per_sample,
                           scale / (float)N, d_x, (long)lddx, n_valid, nemo_red_take(nemo_cdiv(N, 16), 1));
        NEMO_LAUNCH_CHECK();
        return NEMO_OK;
    }
    dim3 grid(nemo_cdiv(N, GMM_TS), (unsigned)M);
    hipLaunchKernelGGL(gmm_ll_kernel<69>, grid, dim3(GMM_TS), 0, (hipStream_t)stream, (long)N, (int)M, x,
                       (long)ldx, means, precisions, log_nllw, ws);
    NEMO_LAUNCH_CHECK();
    hipLaunchKernelGGL(gmm_grad_kernel<69>, grid, dim3(GMM_TS), 0, (hipStream_t)stream, (long)N, (int)M, x,
                       (long)ldx, means, precisions, ws, scalar_out, per_sample, scale / (float)N, d_x,
                       (long)lddx, n_valid, nemo_red_take(grid.x, 1));
    NEMO_LAUNCH_CHECK();
    return NEMO_OK;
}

extern "C" int32_t nemo_pose3d_fwd_bwd(int64_t N, int64_t dim, const float* x, int64_t ldx,
                                       const float* target, const float* mask, const int64_t* view_idx,
                                       const int64_t* frame_idx, int64_t T, float* scalar_out, float scale,
                                       float* d_x, int64_t lddx, const int64_t* n_valid, void* stream) {
    if (N <= 0 || dim <= 0 || !x || !target || !mask || !view_idx || !frame_idx || !scalar_out) return NEMO_EINVAL;
    int blocks = nemo_cdiv(N * dim, 256);
    if (blocks > 1024) blocks = 1024;
    hipLaunchKernelGGL(pose3d_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (long)N, (int)dim, x,
                       (long)ldx, target, mask, view_idx, frame_idx, (long)T, scalar_out, scale, d_x,
                       (long)lddx, n_valid, nemo_red_take(blocks, 1));
    NEMO_LAUNCH_CHECK();
    return NEMO_OK;
}

extern "C" int32_t nemo_adam_step(int32_t n_seg, const nemo_adam_seg* segs, float* params, const float* grads,
                                  float* exp_avg, float* exp_avg_sq, float beta1, float beta2, float eps,
                                  void* stream) {
    if (n_seg < 0 || n_seg > NEMO_ADAM_MAX_SEG || !segs || !params || !grads || !exp_avg || !exp_avg_sq)
        return NEMO_EINVAL;
    if (n_seg == 0) return NEMO_OK;
    AdamSegs a;
    a.n = n_seg;
    long maxn = 0;
    for (int i = 0; i < n_seg; ++i) {
        a.s[i] = segs[i];
        if (segs[i].numel > maxn) maxn = segs[i].numel;
    }
    int bx = nemo_cdiv(maxn, 256 * 4);
    if (bx < 1) bx = 1;
    if (bx > 2048) bx = 2048;
    hipLaunchKernelGGL(adam_kernel<false>, dim3(bx, n_seg), dim3(256), 0, (hipStream_t)stream, a, nullptr,
                       params, grads, exp_avg, exp_avg_sq, beta1, beta2, eps, (const float*)nullptr);
    NEMO_LAUNCH_CHECK();
    return NEMO_OK;
}

extern "C" int32_t nemo_adam_step_dev_if(int32_t n_seg, const nemo_adam_seg* segs_dev, int64_t max_numel,
                                         float* params, const float* grads, float* exp_avg, float* exp_avg_sq,
                                         float beta1, float beta2, float eps, const float* skip_if_nonzero, void* stream) {
    if (n_seg < 0 || n_seg > NEMO_ADAM_MAX_SEG || !segs_dev || !params || !grads || !exp_avg || !exp_avg_sq ||
        max_numel < 0)
        return NEMO_EINVAL;
    if (n_seg == 0 || max_numel == 0) return NEMO_OK;
    AdamSegs a;
    a.n = n_seg;
    int bx = nemo_cdiv(max_numel, 256 * 4);
    if (bx < 1) bx = 1;
    if (bx > 2048) bx = 2048;
    hipLaunchKernelGGL(adam_kernel<true>, dim3(bx, n_seg), dim3(256), 0, (hipStream_t)stream, a, segs_dev,
                       params, grads, exp_avg, exp_avg_sq, beta1, beta2, eps, skip_if_nonzero);
    NEMO_LAUNCH_CHECK();
    return NEMO_OK;
}

extern "C" int32_t nemo_adam_step_dev(int32_t n_seg, const nemo_adam_seg* segs_dev, int64_t max_numel,
                                      float* params, const float* grads, float* exp_avg, float* exp_avg_sq,
                                      float beta1, float beta2, float eps, void* stream) {
    return nemo_adam_step_dev_if(n_seg, segs_dev, max_numel, params, grads, exp_avg, exp_avg_sq, beta1, beta2, eps, nullptr,
                                 stream);
}

namespace {
__global__ __launch_bounds__(256) void nan_count_kernel(const float* __restrict__ x, long n, float* __restrict__ out) {
    __shared__ float red[16];
    float c = 0.f;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) c += x[i] != x[i] ? 1.f : 0.f;
    const float t = block_sum(c, red);
    if (threadIdx.x == 0 && t != 0.f) atomicAdd(out, t);
}

__global__ __launch_bounds__(256) void seq_gather_kernel(const long* __restrict__ all_v, const long* __restrict__ all_f, long B,
                                                         const int* __restrict__ counter, long* __restrict__ vo,
                                                         long* __restrict__ fo) {
    const long row = (long)*counter * B;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < B; i += (long)gridDim.x * blockDim.x) {
        vo[i] = all_v[row + i];
        fo[i] = all_f[row + i];
    }
}

__global__ void seq_log_kernel(const float* __restrict__ src, int n, float* __restrict__ log, long ld, int* __restrict__ counter) {
    const int c = *counter;
    if ((int)threadIdx.x < n) log[(long)c * ld + threadIdx.x] = src[threadIdx.x];
    __syncthreads();
    if (threadIdx.x == 0) *counter = c + 1;
}
}  // namespace

extern "C" int32_t nemo_nan_count(const float* x, int64_t n, float* count_out, void* stream) {
    if (n < 0 || (n && !x) || !count_out) return NEMO_EINVAL;
    if (n == 0) return NEMO_OK;
    int blocks = nemo_cdiv(n, 256 * 8);
    if (blocks > 1024) blocks = 1024;
    hipLaunchKernelGGL(nan_count_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, x, (long)n, count_out);
    NEMO_LAUNCH_CHECK();
    return NEMO_OK;
}

extern "C" int32_t nemo_seq_gather(const int64_t* all_view, const int64_t* all_frame, int64_t B, const int32_t* counter,
                                   int64_t* view_out, int64_t* frame_out, void* stream) {
    if (B < 0 || !all_view || !all_frame || !counter || !view_out || !frame_out) return NEMO_EINVAL;
    if (B == 0) return NEMO_OK;
    int blocks = nemo_cdiv(B, 256);
    if (blocks > 64) blocks = 64;
    hipLaunchKernelGGL(seq_gather_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const long*)all_view,
                       (const long*)all_frame, (long)B, (const int*)counter, (long*)view_out, (long*)frame_out);
    NEMO_LAUNCH_CHECK();
    return NEMO_OK;
}

extern "C" int32_t nemo_seq_log(const float* src, int32_t n, float* log, int64_t ld, int32_t* counter, void* stream) {
    if (!src || !log || !counter || n < 1 || n > 64 || ld < n) return NEMO_EINVAL;
    hipLaunchKernelGGL(seq_log_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, src, (int)n, log, (long)ld, (int*)counter);
    NEMO_LAUNCH_CHECK();
    return NEMO_OK;
}
