// fp32 GEMM on the CDNA4 matrix cores (v_mfma_f32_32x32x2_f32), LDS-tiled, with the fused
// epilogues the NeMo step needs (bias, ReLU / LeakyReLU, activation-gradient masks,
// accumulate, split-K atomics).  Exact fp32 arithmetic (the MFMA is a k-ordered fmaf chain),
// which is what the 1e-4 parity gate of the fit requires.
//
// Replaces, on the hot path of the reference: nn.Linear in FCNN / MotionNet
// (nemo/neural_motion_model.py:58-71,130-148), VPoser's Linear layers
// (human_body_prior/models/vposer_model.py:69-88), the pose-blend contraction
// (human_body_prior/body_model/lbs.py:229-233) and their autograd backward GEMMs.
//
// Structure: 256 threads = 4 waves in a 2x2 grid, each wave owns (BM/2)x(BN/2) of the block tile as
// 32x32 MFMA accumulators.  K is walked in tiles of 16 through a double-buffered LDS image stored
// k-major ([k][m], [k][n]) so that the per-lane MFMA operand reads are conflict-free ds_read_b32;
// one barrier per K-tile; the next tile's global loads are issued before the MFMAs of the current
// one.  Operands whose rows are 16-byte aligned are staged with dwordx4 loads (VEC path), anything
// else (e.g. nn.Linear weights with in_features = 105 / 63) with dword loads.
#include "common.h"
#include "../../include/nemo_hip.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace {

struct GemmArgs {
    const float* A; const float* B; float* C;
    const float* bias; const float* mask;
    long M, N, K, lda, ldb, ldc, ldmask;
    long k_chunk;           // K range per grid.z slice (multiple of the K tile)
    int tiles_m, tiles_n;
    int act, mask_mode, out_mode;
    float alpha;
};

constexpr int PAD = 4;

// Per-thread staging of (ROWS x BKT) operand tiles, two tiles in flight (register slots 0/1).
// Element (r, kk) of a tile is
//   KCONTIG : src[(row0 + r) * ld + k0 + kk]      (k is the contiguous axis)
//   !KCONTIG: src[(k0 + kk) * ld + row0 + r]      (r is the contiguous axis)
// Out-of-range elements are zero.
template <int ROWS, int BKT, bool KCONTIG, bool VEC>
struct Stager {
    static constexpr int NV = VEC ? ROWS * BKT / 1024 : ROWS * BKT / 256;   // float4 / float per thread per tile
    const float* p[NV];           // per-element pointers (advanced every K tile)
    bool row_ok[NV];
    int r_[NV], k_[NV];           // element position inside the tile
    float4 v4[2][VEC ? NV : 1];
    float v1[2][VEC ? 1 : NV];
    long step;                    // pointer advance per K tile
    const float* safe;            // always-valid, 16-byte aligned address (the operand base)
    int row_rem[NV];              // !KCONTIG VEC: rows remaining from the element's first row (clamped)

    __device__ __forceinline__ void init(const float* src, long ld, long row0, long k0, long row_lim) {
        const int t = threadIdx.x;
        safe = src;
        step = KCONTIG ? BKT : (long)BKT * ld;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            int r, kk;
            if (VEC) {
                if (KCONTIG) { kk = 4 * (t % (BKT / 4)); r = t / (BKT / 4) + (1024 / BKT) * i; }
                else { r = 4 * (t % (ROWS / 4)); kk = t / (ROWS / 4) + (1024 / ROWS) * i; }
            } else {
                if (KCONTIG) { kk = t % BKT; r = t / BKT + (256 / BKT) * i; }
                else { r = t % ROWS; kk = t / ROWS + (256 / ROWS) * i; }
            }
            r_[i] = r; k_[i] = kk;
            const long gr = row0 + r;
            const long rem = row_lim - gr;
            row_rem[i] = rem > 4 ? 4 : (rem < 0 ? 0 : (int)rem);
            row_ok[i] = (VEC && !KCONTIG) ? (gr + 3 < row_lim) : (gr < row_lim);
            p[i] = KCONTIG ? src + gr * ld + k0 + kk : src + (k0 + kk) * ld + gr;
        }
    }

    // k_rem = k_lim - k0 of the tile being loaded (may be <= 0: zero fill).
    // BRANCH-FREE: every load is unconditional from a clamped, always-valid address and invalid
    // elements are zeroed with selects.  (A predicated load sits in its own basic block and hipcc
    // waits for it at the block end -- that serialised every staging load behind a full L2 round
    // trip.)  A dwordx4 that straddles the logical end of a row stays inside the row's storage because
    // the VEC path requires ld % 4 == 0 (so ld >= the extent rounded up to 4).
    template <int SLOT>
    __device__ __forceinline__ void load(long k_rem) {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            if (VEC) {
                int nvalid;                       // leading valid elements of the float4 (0..4)
                if (KCONTIG) {
                    const long kr = k_rem - k_[i];
                    nvalid = row_ok[i] ? (kr > 4 ? 4 : (kr < 0 ? 0 : (int)kr)) : 0;
                } else {
                    nvalid = (k_[i] < k_rem) ? row_rem[i] : 0;
                }
                const float* q = nvalid > 0 ? p[i] : safe;
                float4 v = *reinterpret_cast<const float4*>(q);
                v.x = nvalid > 0 ? v.x : 0.f; v.y = nvalid > 1 ? v.y : 0.f;
                v.z = nvalid > 2 ? v.z : 0.f; v.w = nvalid > 3 ? v.w : 0.f;
                v4[SLOT][i] = v;
            } else {
                const bool ok = row_ok[i] && k_[i] < k_rem;
                const float* q = ok ? p[i] : safe;
                const float v = *q;
                v1[SLOT][i] = ok ? v : 0.f;
            }
            p[i] += step;
        }
    }

    template <int SLOT>
    __device__ __forceinline__ void store(float (*lds)[ROWS + PAD]) const {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            if (VEC) {
                if (KCONTIG) {
                    lds[k_[i] + 0][r_[i]] = v4[SLOT][i].x; lds[k_[i] + 1][r_[i]] = v4[SLOT][i].y;
                    lds[k_[i] + 2][r_[i]] = v4[SLOT][i].z; lds[k_[i] + 3][r_[i]] = v4[SLOT][i].w;
                } else {
                    *reinterpret_cast<float4*>(&lds[k_[i]][r_[i]]) = v4[SLOT][i];
                }
            } else {
                lds[k_[i]][r_[i]] = v1[SLOT][i];
            }
        }
    }
};

template <int BM, int BN, int BKT, bool TA, bool TB, bool VEC>
__global__ __launch_bounds__(256) void gemm_f32_kernel(GemmArgs g) {
    constexpr int WM = BM / 2, WN = BN / 2, TM = WM / 32, TN = WN / 32;
    __shared__ __attribute__((aligned(16))) float As[2][BKT][BM + PAD];
    __shared__ __attribute__((aligned(16))) float Bs[2][BKT][BN + PAD];

    const int bid = blockIdx.x;
    const int tm = bid % g.tiles_m, tn = bid / g.tiles_m;
    const long m0 = (long)tm * BM, n0 = (long)tn * BN;
    const long kbeg = (long)blockIdx.z * g.k_chunk;
    const long kend = min(g.K, kbeg + g.k_chunk);

    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int wm = wid >> 1, wn = wid & 1;
    const int l31 = lane & 31, lhi = lane >> 5;

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // A tile: "rows" = m; transA == 0 -> A[m][k], k contiguous.  B tile: "rows" = n; transB == 1 -> B[n][k].
    Stager<BM, BKT, !TA, VEC> sa;
    Stager<BN, BKT, TB, VEC> sb;
    sa.init(g.A, g.lda, m0, kbeg, g.M);
    sb.init(g.B, g.ldb, n0, kbeg, g.N);

    auto compute = [&](int cur) {
        if constexpr (TM * TN == 1) {
            // one MFMA per k-step cannot cover the LDS latency of its own operands: fetch the whole
            // tile's operands first (BKT ds_reads in flight), then issue the MFMAs back to back
            float a[BKT / 2], b[BKT / 2];
#pragma unroll
            for (int ks = 0; ks < BKT; ks += 2) {
                a[ks / 2] = As[cur][ks + lhi][wm * WM + l31];
                b[ks / 2] = Bs[cur][ks + lhi][wn * WN + l31];
            }
#pragma unroll
            for (int ks = 0; ks < BKT / 2; ++ks)
                acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[ks], b[ks], acc[0][0], 0, 0, 0);
        } else {
#pragma unroll
            for (int ks = 0; ks < BKT; ks += 2) {
                float a[TM], b[TN];
#pragma unroll
                for (int i = 0; i < TM; ++i) a[i] = As[cur][ks + lhi][wm * WM + i * 32 + l31];
#pragma unroll
                for (int j = 0; j < TN; ++j) b[j] = Bs[cur][ks + lhi][wn * WN + j * 32 + l31];
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
            }
        }
    };

    if constexpr (BM == 64) {
        // Software pipeline, two K tiles in flight in registers: tile t+2 is requested from L2 before the
        // MFMAs of tile t, and lands in LDS (other buffer) only after the MFMAs of tile t+1 -- the global
        // latency is covered by two compute phases even when a CU holds a single block (the small
        // GEMMs of the step put ~1 block on a CU).
        long krem = kend - kbeg;                       // remaining K from the next tile to LOAD
        sa.template load<0>(krem); sb.template load<0>(krem); krem -= BKT;
        sa.template load<1>(krem); sb.template load<1>(krem); krem -= BKT;
        sa.template store<0>(As[0]); sb.template store<0>(Bs[0]);
        __syncthreads();
        for (long k0 = kbeg; k0 < kend; k0 += 2 * BKT) {
            // even tile (LDS buffer 0); register slot 0 is free again
            sa.template load<0>(krem); sb.template load<0>(krem); krem -= BKT;
            compute(0);
            sa.template store<1>(As[1]); sb.template store<1>(Bs[1]);
            __syncthreads();
            // odd tile (LDS buffer 1): a zero tile when K is exhausted (adds nothing)
            sa.template load<1>(krem); sb.template load<1>(krem); krem -= BKT;
            compute(1);
            sa.template store<0>(As[0]); sb.template store<0>(Bs[0]);
            __syncthreads();
        }
    } else {
        // big tiles: 2048 MFMA cycles per K tile already cover the L2 latency; one tile in flight
        if (kbeg < kend) {
            sa.template load<0>(kend - kbeg); sb.template load<0>(kend - kbeg);
            sa.template store<0>(As[0]); sb.template store<0>(Bs[0]);
        }
        __syncthreads();
        int cur = 0;
        for (long k0 = kbeg; k0 < kend; k0 += BKT) {
            const bool more = k0 + BKT < kend;
            if (more) { sa.template load<0>(kend - k0 - BKT); sb.template load<0>(kend - k0 - BKT); }
            compute(cur);
            if (more) { sa.template store<0>(As[cur ^ 1]); sb.template store<0>(Bs[cur ^ 1]); }
            __syncthreads();
            cur ^= 1;
        }
    }

    // Epilogue.  C/D layout of the 32x32 MFMA: col = lane & 31, row = (r&3) + 8*(r>>2) + 4*(lane>>5).
    const bool add_bias = g.bias != nullptr && blockIdx.z == 0;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const long n = n0 + wn * WN + j * 32 + l31;
            if (n >= g.N) continue;
            const float bv = add_bias ? g.bias[n] : 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const long m = m0 + wm * WM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
                if (m >= g.M) continue;
                float v = g.alpha * acc[i][j][r] + bv;
                if (g.act == 1) v = v > 0.f ? v : 0.f;
                else if (g.act == 2) v = v > 0.f ? v : 0.01f * v;
                if (g.mask_mode) {
                    const float mv = g.mask[m * g.ldmask + n];
                    if (g.mask_mode == 1) v = mv > 0.f ? v : 0.f;
                    else v = mv > 0.f ? v : 0.01f * v;
                }
                float* c = g.C + m * g.ldc + n;
                if (g.out_mode == 0) *c = v;
                else if (g.out_mode == 1) *c += v;
                else atomicAdd(c, v);
            }
        }
}

template <int BM, int BN, int BKT, bool VEC>
void launch(int ta, int tb, const GemmArgs& g, dim3 grid, hipStream_t s) {
    if (!ta && !tb) hipLaunchKernelGGL((gemm_f32_kernel<BM, BN, BKT, false, false, VEC>), grid, dim3(256), 0, s, g);
    else if (!ta && tb) hipLaunchKernelGGL((gemm_f32_kernel<BM, BN, BKT, false, true, VEC>), grid, dim3(256), 0, s, g);
    else if (ta && !tb) hipLaunchKernelGGL((gemm_f32_kernel<BM, BN, BKT, true, false, VEC>), grid, dim3(256), 0, s, g);
    else hipLaunchKernelGGL((gemm_f32_kernel<BM, BN, BKT, true, true, VEC>), grid, dim3(256), 0, s, g);
}

constexpr int BK_BIG = 16, BK_SMALL = 32;

inline bool aligned16(const void* p, long ld) { return (((uintptr_t)p) & 15) == 0 && (ld & 3) == 0; }

}  // namespace

extern "C" int32_t nemo_gemm_f32(int32_t transA, int32_t transB, int64_t M, int64_t N, int64_t K,
                                 const float* A, int64_t lda, const float* B, int64_t ldb,
                                 float* C, int64_t ldc, const float* bias, int32_t act,
                                 const float* mask, int64_t ldmask, int32_t mask_mode, float alpha,
                                 int32_t out_mode, int32_t split_k, void* stream) {
    if (M < 0 || N < 0 || K < 0 || !C) return NEMO_EINVAL;
    if (M == 0 || N == 0) return NEMO_OK;
    if (K > 0 && (!A || !B)) return NEMO_EINVAL;
    if (act < 0 || act > 2 || mask_mode < 0 || mask_mode > 2 || out_mode < 0 || out_mode > 2)
        return NEMO_EINVAL;
    if (mask_mode && !mask) return NEMO_EINVAL;
    if (split_k < 1) split_k = 1;
    // a K-split is only linear: no activation / mask, and the partial sums must be added atomically
    if (split_k > 1 && (act || mask_mode || out_mode != 2)) return NEMO_EINVAL;

    GemmArgs g;
    g.A = A; g.B = B; g.C = C; g.bias = bias; g.mask = mask;
    g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldb = ldb; g.ldc = ldc; g.ldmask = ldmask;
    g.act = act; g.mask_mode = mask_mode; g.out_mode = out_mode; g.alpha = alpha;
    // 128x128 tiles (BK 16) once they alone fill the 256 CUs a couple of times over; 64x64 (BK 32) otherwise
    const long big_tiles = ((M + 127) / 128) * ((N + 127) / 128) * split_k;
    const bool big = big_tiles >= 512;
    const int bk = big ? BK_BIG : BK_SMALL;
    long kc = (K + split_k - 1) / split_k;
    kc = ((kc + bk - 1) / bk) * bk;
    if (kc == 0) kc = bk;
    g.k_chunk = kc;
    int nz = (int)((K + kc - 1) / kc);
    if (nz < 1) nz = 1;

    // K-chunk starts are multiples of 16, so operand alignment only depends on the base and the ld
    const bool vec = aligned16(A, lda) && aligned16(B, ldb);
    hipStream_t s = (hipStream_t)stream;
    if (big) {
        g.tiles_m = (int)((M + 127) / 128); g.tiles_n = (int)((N + 127) / 128);
        dim3 grid(g.tiles_m * g.tiles_n, 1, nz);
        if (vec) launch<128, 128, BK_BIG, true>(transA, transB, g, grid, s);
        else launch<128, 128, BK_BIG, false>(transA, transB, g, grid, s);
    } else {
        g.tiles_m = (int)((M + 63) / 64); g.tiles_n = (int)((N + 63) / 64);
        dim3 grid(g.tiles_m * g.tiles_n, 1, nz);
        if (vec) launch<64, 64, BK_SMALL, true>(transA, transB, g, grid, s);
        else launch<64, 64, BK_SMALL, false>(transA, transB, g, grid, s);
    }
    NEMO_LAUNCH_CHECK();
    return NEMO_OK;
}

// Column sums of a row-major (M x N) matrix: out[n] (+)= sum_m X[m][n].  Bias gradients.
namespace {
__global__ __launch_bounds__(256) void colsum_kernel(const float* __restrict__ X, long M, long N, long ldx,
                                                     float* __restrict__ out, long rows_per_block) {
    const long n = (long)blockIdx.x * 64 + (threadIdx.x & 63);
    const long mbeg = (long)blockIdx.y * rows_per_block;
    const long mend = min(M, mbeg + rows_per_block);
    float s = 0.f;
    if (n < N)
        for (long m = mbeg + (threadIdx.x >> 6); m < mend; m += 4) s += X[m * ldx + n];
    __shared__ float red[4][64];
    red[threadIdx.x >> 6][threadIdx.x & 63] = s;
    __syncthreads();
    if (threadIdx.x < 64 && n < N) {
        const float t = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
        atomicAdd(out + n, t);
    }
}
}  // namespace

namespace {
struct ColsumBatch { nemo_colsum_desc d[NEMO_COLSUM_MAX]; };
__global__ __launch_bounds__(256) void colsum_multi_kernel(ColsumBatch b, long rows_per_block) {
    const nemo_colsum_desc d = b.d[blockIdx.z];
    const long n = (long)blockIdx.x * 64 + (threadIdx.x & 63);
    const long mbeg = (long)blockIdx.y * rows_per_block;
    if (mbeg >= d.M || (long)blockIdx.x * 64 >= d.N) return;       // block-uniform
    const long mend = min((long)d.M, mbeg + rows_per_block);
    float s = 0.f;
    if (n < d.N)
        for (long m = mbeg + (threadIdx.x >> 6); m < mend; m += 4) s += d.X[m * d.ldx + n];
    __shared__ float red[4][64];
    red[threadIdx.x >> 6][threadIdx.x & 63] = s;
    __syncthreads();
    if (threadIdx.x < 64 && n < d.N) {
        const float t = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
        atomicAdd(d.out + n, t);
    }
}
}  // namespace

extern "C" int32_t nemo_colsum_multi(int32_t n, const nemo_colsum_desc* descs, void* stream) {
    if (n < 0 || n > NEMO_COLSUM_MAX || (n && !descs)) return NEMO_EINVAL;
    if (n == 0) return NEMO_OK;
    ColsumBatch b;
    long maxM = 0, maxN = 0;
    for (int i = 0; i < n; ++i) {
        if (descs[i].M < 0 || descs[i].N < 0 || !descs[i].X || !descs[i].out) return NEMO_EINVAL;
        b.d[i] = descs[i];
        if (descs[i].M > maxM) maxM = descs[i].M;
        if (descs[i].N > maxN) maxN = descs[i].N;
    }
    if (maxM == 0 || maxN == 0) return NEMO_OK;
    const long rows_per_block = 128;
    dim3 grid(nemo_cdiv(maxN, 64), nemo_cdiv(maxM, rows_per_block), n);
    hipLaunchKernelGGL(colsum_multi_kernel, grid, dim3(256), 0, (hipStream_t)stream, b, rows_per_block);
    NEMO_LAUNCH_CHECK();
    return NEMO_OK;
}

extern "C" int32_t nemo_colsum_f32(const float* X, int64_t M, int64_t N, int64_t ldx, float* out,
                                   void* stream) {
    if (M < 0 || N < 0 || !out) return NEMO_EINVAL;
    if (M == 0 || N == 0) return NEMO_OK;
    if (!X) return NEMO_EINVAL;
    const long rows_per_block = 256;
    dim3 grid(nemo_cdiv(N, 64), nemo_cdiv(M, rows_per_block));
    hipLaunchKernelGGL(colsum_kernel, grid, dim3(256), 0, (hipStream_t)stream, X, (long)M, (long)N,
                       (long)ldx, out, rows_per_block);
    NEMO_LAUNCH_CHECK();
    return NEMO_OK;
}
