// fp32 GEMM on the CDNA4 matrix cores (v_mfma_f32_32x32x2_f32), LDS-tiled, with the fused
// epilogues the NeMo step needs (bias, ReLU / LeakyReLU, activation-gradient masks,
// accumulate, split-K atomics).  Exact fp32 arithmetic (the MFMA is a k-ordered fmaf chain),
// which is what the 1e-4 parity gate of the fit requires.
//
// Replaces, on the hot path of the reference: nn.Linear in FCNN / MotionNet
// (nemo/neural_motion_model.py:58-71,130-148), VPoser's Linear layers
// (human_body_prior/models/vposer_model.py:69-88), the pose-blend contraction
// (human_body_prior/body_model/lbs.py:229-233) and their autograd backward GEMMs.
#include "common.h"
#include "../../include/nemo_hip.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace {

struct GemmArgs {
    const float* A; const float* B; float* C;
    const float* bias; const float* mask;
    long M, N, K, lda, ldb, ldc, ldmask;
    long k_chunk;           // K range per grid.z slice (multiple of BK)
    int tiles_m, tiles_n;
    int act, mask_mode, out_mode;
    float alpha;
};

constexpr int BK = 16;
constexpr int PAD = 4;

// Load a (ROWS x BK) operand tile into registers.  Element (r, kk) of the tile is
//   KCONTIG : src[(row0 + r) * ld + k0 + kk]      (k is the contiguous axis)
//   !KCONTIG: src[(k0 + kk) * ld + row0 + r]      (r is the contiguous axis)
// Out-of-range elements are zero.  Thread->element mapping keeps global reads coalesced
// along the contiguous axis.
template <int ROWS, bool KCONTIG>
__device__ __forceinline__ void load_tile(const float* __restrict__ src, long ld, long row0, long k0,
                                          long row_lim, long k_lim, float (&reg)[ROWS / 16]) {
    const int t = threadIdx.x;
#pragma unroll
    for (int i = 0; i < ROWS / 16; ++i) {
        int r, kk;
        if (KCONTIG) { kk = t & 15; r = (t >> 4) + 16 * i; }
        else         { r = t % ROWS; kk = t / ROWS + (256 / ROWS) * i; }
        const long gr = row0 + r, gk = k0 + kk;
        float v = 0.f;
        if (gr < row_lim && gk < k_lim) v = KCONTIG ? src[gr * ld + gk] : src[gk * ld + gr];
        reg[i] = v;
    }
}

template <int ROWS, bool KCONTIG>
__device__ __forceinline__ void store_tile(float (*lds)[ROWS + PAD], const float (&reg)[ROWS / 16]) {
    const int t = threadIdx.x;
#pragma unroll
    for (int i = 0; i < ROWS / 16; ++i) {
        int r, kk;
        if (KCONTIG) { kk = t & 15; r = (t >> 4) + 16 * i; }
        else         { r = t % ROWS; kk = t / ROWS + (256 / ROWS) * i; }
        lds[kk][r] = reg[i];
    }
}

template <int BM, int BN, bool TA, bool TB>
__global__ __launch_bounds__(256) void gemm_f32_kernel(GemmArgs g) {
    constexpr int WM = BM / 2, WN = BN / 2, TM = WM / 32, TN = WN / 32;
    __shared__ float As[BK][BM + PAD];
    __shared__ float Bs[BK][BN + PAD];

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

    float ra[BM / 16], rb[BN / 16];
    // A tile: "rows" = m.  transA == 0 -> A[m][k], k contiguous.
    load_tile<BM, !TA>(g.A, g.lda, m0, kbeg, g.M, kend, ra);
    // B tile: "rows" = n.  transB == 1 -> B[n][k], k contiguous.
    load_tile<BN, TB>(g.B, g.ldb, n0, kbeg, g.N, kend, rb);

    for (long k0 = kbeg; k0 < kend; k0 += BK) {
        __syncthreads();   // previous iteration's LDS reads are done
        store_tile<BM, !TA>(As, ra);
        store_tile<BN, TB>(Bs, rb);
        __syncthreads();
        if (k0 + BK < kend) {   // prefetch the next tile while the MFMAs run
            load_tile<BM, !TA>(g.A, g.lda, m0, k0 + BK, g.M, kend, ra);
            load_tile<BN, TB>(g.B, g.ldb, n0, k0 + BK, g.N, kend, rb);
        }
#pragma unroll
        for (int ks = 0; ks < BK; ks += 2) {
            float a[TM], b[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) a[i] = As[ks + lhi][wm * WM + i * 32 + l31];
#pragma unroll
            for (int j = 0; j < TN; ++j) b[j] = Bs[ks + lhi][wn * WN + j * 32 + l31];
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
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

template <int BM, int BN>
void launch(int ta, int tb, const GemmArgs& g, dim3 grid, hipStream_t s) {
    if (!ta && !tb) hipLaunchKernelGGL((gemm_f32_kernel<BM, BN, false, false>), grid, dim3(256), 0, s, g);
    else if (!ta && tb) hipLaunchKernelGGL((gemm_f32_kernel<BM, BN, false, true>), grid, dim3(256), 0, s, g);
    else if (ta && !tb) hipLaunchKernelGGL((gemm_f32_kernel<BM, BN, true, false>), grid, dim3(256), 0, s, g);
    else hipLaunchKernelGGL((gemm_f32_kernel<BM, BN, true, true>), grid, dim3(256), 0, s, g);
}

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
    long kc = (K + split_k - 1) / split_k;
    kc = ((kc + BK - 1) / BK) * BK;
    if (kc == 0) kc = BK;
    g.k_chunk = kc;
    int nz = (int)((K + kc - 1) / kc);
    if (nz < 1) nz = 1;

    // 128x128 tiles once they alone fill the 256 CUs a couple of times over; 64x64 otherwise
    const long big_tiles = ((M + 127) / 128) * ((N + 127) / 128) * nz;
    hipStream_t s = (hipStream_t)stream;
    if (big_tiles >= 512) {
        g.tiles_m = (int)((M + 127) / 128); g.tiles_n = (int)((N + 127) / 128);
        launch<128, 128>(transA, transB, g, dim3(g.tiles_m * g.tiles_n, 1, nz), s);
    } else {
        g.tiles_m = (int)((M + 63) / 64); g.tiles_n = (int)((N + 63) / 64);
        launch<64, 64>(transA, transB, g, dim3(g.tiles_m * g.tiles_n, 1, nz), s);
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
