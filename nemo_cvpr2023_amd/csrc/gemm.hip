// fp32 GEMM on the CDNA4 matrix cores (v_mfma_f32_32x32x2_f32), LDS-tiled, with the fused
// epilogues the NeMo step needs (bias, ReLU / LeakyReLU, activation-gradient masks, accumulate)
// and an in-launch split-K whose partial sums are combined by the last-arriving block in a fixed
// order (deterministic, epilogue still fused).  Exact fp32 arithmetic (the MFMA is an fmaf chain),
// which is what the 1e-4 parity gate of the fit requires.
//
// Replaces, on the hot path of the reference: nn.Linear in FCNN / MotionNet
// (nemo/neural_motion_model.py:58-71,130-148), VPoser's Linear layers
// (human_body_prior/models/vposer_model.py:69-88), the pose-blend contraction
// (human_body_prior/body_model/lbs.py:229-233) and their autograd backward GEMMs.
//
// Structure: 256 threads = 4 waves in a 2x2 grid, each wave owns (BM/2)x(BN/2) of the block tile as
// 32x32 MFMA accumulators (128x128 tile: 4 per wave, 64x64: 1).  K is walked in tiles of 32 through a
// double-buffered LDS image; one barrier per K tile; the next tile's global loads are issued before
// the MFMAs of the current one and land in LDS just before its last MFMA group.
//
// k permutation: MFMA step j of k-group q takes k = 8q + j from lanes 0-31 and k = 8q + 4 + j from
// lanes 32-63 (for both operands, so the dot product is complete -- only the fp32 summation order
// differs from a sequential k walk).  A lane's four operands of a group are then CONTIGUOUS in k:
// an operand whose source is k-contiguous is staged row-major ([row][k], dwordx4 global loads ->
// ds_write_b128) and fetched with ONE conflict-free ds_read_b128 per group (row stride 36 floats: the
// 16-lane service groups of a b128 read cover all 64 banks); an operand whose source is
// row-contiguous is staged k-major ([k][row]) and fetched with four ds_read_b32.
#include <type_traits>
#include <cstdio>
#include <cstdlib>
#include "common.h"
#include "gemm_glds.h"
#include "gemm_adj.h"
#include "gemm_skinny.h"
#include "gemm_b16x.h"
#include "gemm_xp.h"
#include "../../include/nemo_hip.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4v __attribute__((ext_vector_type(4)));

namespace {

// (shared with the LDS-DMA generation of the kernel, gemm_glds.h: same tile / slice bookkeeping, same slab layout)
using GemmArgs = glds::Args;

// Per-thread staging of one (ROWS x BK) operand tile.  Element (r, kk) of a tile is
//   KCONTIG : src[(row0 + r) * ld + k0 + kk]      (k is the contiguous axis)  -> LDS [r][BK+4]
//   !KCONTIG: src[(k0 + kk) * ld + row0 + r]      (r is the contiguous axis)  -> LDS [kk][ROWS+4]
// Rows beyond the operand are NOT zeroed: they only feed output rows / columns the epilogue never
// writes, so their loads are simply redirected to a valid row (pointer fixed once, in init).  Only
// k >= K must contribute zeros; that concerns the last K tile alone, which goes through the MASKED
// variants.  The full-tile variants are one unconditional load and one unconditional LDS store per
// element -- nothing that could make hipcc wait for a load before the MFMAs it is meant to hide
// behind (a predicated load, or a select on the loaded value, puts the s_waitcnt right there).
template <int ROWS, int BK, bool KCONTIG, bool VEC, int NS>
struct Stager {
    static constexpr int NV = VEC ? ROWS * BK / 1024 : ROWS * BK / 256;   // float4 / float per thread per tile
    // LDS image: [row][BK+4] (k-contiguous: ONE ds_read_b128 per operand group) for k-contiguous sources and,
    // TRANSPOSED ON THE WAY IN, for row-contiguous sources staged with dwordx4 (64-row tiles): a wave's
    // load covers 4 row chunks x 16 k rows, a half-wave two chunks (one even, one odd) x 16 k, so the four
    // ds_write_b32 of a float4 (rows 4c..4c+3, one k) hit banks (16 c + 4 j + k) mod 32 -- 32 distinct per half-wave.  Row-contiguous sources otherwise stay k-major ([k][ROWS+4],
    // four ds_read_b32 per group: the parameter-gradient GEMMs ran at ~0.9 of the NT rate's time per K tile).
    static constexpr bool TRANSP = !KCONTIG && VEC && ROWS == 64 && BK == 32;
    static constexpr bool LK = KCONTIG || TRANSP;                         // LDS image is k-contiguous
    static constexpr int LD = LK ? BK + 4 : ROWS + 4;
    static constexpr int SIZE = LK ? ROWS * LD : BK * LD;
    const float* p[NV];           // per-element pointers (advanced every K tile)
    int off_[NV], k_[NV];         // LDS offset / k position inside the tile
    float4 v4[NS][VEC ? NV : 1];  // NS register slots = NS K tiles in flight
    float v1[NS][VEC ? 1 : NV];
    long step;                    // pointer advance per K tile
    const float* safe;            // always-valid, 16-byte aligned address (the operand base)

    __device__ __forceinline__ void init(const float* src, long ld, long row0, long k0, long row_lim) {
        const int t = threadIdx.x;
        safe = src;
        step = KCONTIG ? BK : (long)BK * ld;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            int r, kk;
            if (VEC) {
                if (KCONTIG) { kk = 4 * (t % (BK / 4)); r = t / (BK / 4) + (1024 / BK) * i; }
                else if (TRANSP) { r = 4 * (((t >> 4) & 3) + 4 * (t >> 6)); kk = (t & 15) + 16 * i; }
                else { r = 4 * (t % (ROWS / 4)); kk = t / (ROWS / 4) + (1024 / ROWS) * i; }
            } else {
                if (KCONTIG) { kk = t % BK; r = t / BK + (256 / BK) * i; }
                else { r = t % ROWS; kk = t / ROWS + (256 / ROWS) * i; }
            }
            k_[i] = kk;
            off_[i] = LK ? r * LD + kk : kk * LD + r;
            long gr = row0 + r;
            // a dwordx4 along rows that straddles row_lim stays inside the row storage (ld % 4 == 0)
            if (gr >= row_lim) gr = 0;
            p[i] = KCONTIG ? src + gr * ld + k0 + kk : src + (k0 + kk) * ld + gr;
        }
    }

    // leading valid k elements of element i when k_rem = K_end - k0 of the tile
    __device__ __forceinline__ int nvalid(int i, long k_rem) const {
        const long kr = k_rem - k_[i];
        if (VEC && KCONTIG) return kr > 4 ? 4 : (kr < 0 ? 0 : (int)kr);
        return kr > 0 ? 4 : 0;
    }

    template <bool MASKED, int SLOT>
    __device__ __forceinline__ void load(long k_rem) {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const float* q = p[i];
            if (MASKED) q = nvalid(i, k_rem) > 0 ? q : safe;      // never dereference beyond the last K row
            if (VEC) v4[SLOT][i] = *reinterpret_cast<const float4*>(q);
            else v1[SLOT][i] = *q;
            p[i] += step;
        }
    }

    // k_rem: the same value the matching load() was given
    template <bool MASKED, int SLOT>
    __device__ __forceinline__ void store(float* lds, long k_rem) const {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            if (VEC) {
                float4 v = v4[SLOT][i];
                if (MASKED) {
                    const int nv = nvalid(i, k_rem);
                    v.x = nv > 0 ? v.x : 0.f; v.y = nv > 1 ? v.y : 0.f;
                    v.z = nv > 2 ? v.z : 0.f; v.w = nv > 3 ? v.w : 0.f;
                }
                if (TRANSP) {
                    float* d = lds + off_[i];
                    d[0] = v.x; d[LD] = v.y; d[2 * LD] = v.z; d[3 * LD] = v.w;
                } else {
                    *reinterpret_cast<float4*>(lds + off_[i]) = v;
                }
            } else {
                lds[off_[i]] = (!MASKED || nvalid(i, k_rem) > 0) ? v1[SLOT][i] : 0.f;
            }
        }
    }
};

// the four operands (MFMA steps j = 0..3) of k-group q for the 32 rows starting at row0
template <int ROWS, int BK, bool KCONTIG>
__device__ __forceinline__ void fetch_group(const float* lds, int row, int q, int lhi, float (&f)[4]) {
    if (KCONTIG) {
        const float4 v = *reinterpret_cast<const float4*>(lds + row * (BK + 4) + 8 * q + 4 * lhi);
        f[0] = v.x; f[1] = v.y; f[2] = v.z; f[3] = v.w;
    } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) f[j] = lds[(8 * q + 4 * lhi + j) * (ROWS + 4) + row];
    }
}

template <int BM, int BN, int BK, bool TA, bool TB, bool VA, bool VB, int NS>
__global__ __launch_bounds__(256, 2) void gemm_f32_kernel(GemmArgs g) {
    constexpr int WM = BM / 2, WN = BN / 2, TM = WM / 32, TN = WN / 32, G = BK / 8;
    using SA = Stager<BM, BK, !TA, VA, NS>;
    using SB = Stager<BN, BK, TB, VB, NS>;
    extern __shared__ __attribute__((aligned(16))) float smem[];      // the ONLY LDS object of the kernel
    float* const As0 = smem;
    float* const Bs0 = smem + 2 * SA::SIZE;

    // Block -> (tile, K slice).  The first t0 tiles are whole; only the tail [t0, n_tiles) -- the tiles beyond a
    // multiple of the CU count, which would otherwise cost a full extra round of blocks -- is cut along K
    // (t0 = 0: every tile is cut into `split` slices, the plain split-K launch).
    const int bid = blockIdx.x;
    const bool whole = bid < g.t0;
    const int n_tail = g.n_tiles - g.t0;
    const int tile = whole ? bid : g.t0 + (bid - g.t0) % n_tail;
    const int slice = whole ? 0 : (bid - g.t0) / n_tail;
    const int split = whole ? 1 : g.split;
    const int tm = tile % g.tiles_m, tn = tile / g.tiles_m;
    const long m0 = (long)tm * BM, n0 = (long)tn * BN;
    const long kbeg = whole ? 0 : (long)slice * g.k_chunk;
    const long kend = whole ? g.K : min(g.K, kbeg + g.k_chunk);
    const long klen = kend > kbeg ? kend - kbeg : 0;
    const long nt = (klen + BK - 1) / BK;                 // K tiles of this slice

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
    SA sa;
    SB sb;
    sa.init(g.A, g.lda, m0, kbeg, g.M);
    sb.init(g.B, g.ldb, n0, kbeg, g.N);

    // Register pipeline, NS K tiles in flight.  Iteration t (LDS buffer t&1 holds tile t):
    //   1. tile t+1 leaves its register slot for the OTHER LDS buffer (free since the barrier that ended
    //      iteration t-1) -- at the START of the iteration, so the ds_writes drain under the MFMAs and
    //      the closing barrier never waits for them;
    //   2. tile t+1+NS is requested into the slot just vacated (it has NS iterations of MFMA work to
    //      hide behind, even when the CU holds this block alone);
    //   3. tile t's MFMAs from LDS, operand groups double-buffered in registers;   4. one barrier.
    // Prologue: tile 0 -> LDS buffer 0; tiles 1..NS in flight in slots 1 % NS .. NS % NS.
    sa.template load<true, 0>(klen); sb.template load<true, 0>(klen);
    sa.template store<true, 0>(As0, klen); sb.template store<true, 0>(Bs0, klen);
    __builtin_amdgcn_sched_barrier(0);
    auto preload = [&](auto tile_no) {
        constexpr int J = decltype(tile_no)::value;
        sa.template load<true, J % NS>(klen - J * BK); sb.template load<true, J % NS>(klen - J * BK);
        __builtin_amdgcn_sched_barrier(0);     // keep the request order: the loop's vmcnt counts rely on it
    };
    preload(std::integral_constant<int, 1>{});
    if constexpr (NS > 1) preload(std::integral_constant<int, 2>{});
    if constexpr (NS > 2) preload(std::integral_constant<int, 3>{});
    if constexpr (NS > 3) preload(std::integral_constant<int, 4>{});
    __syncthreads();
    int cur = 0;
    // MASKED = the tiles staged may be partial or absent (zero fill); the steady state is unmasked.
    auto k_tile = [&](auto masked, auto uslot, long t) {
        constexpr bool MASKED = decltype(masked)::value;
        constexpr int U = decltype(uslot)::value;             // = (t + 1) % NS
        const float* as = As0 + cur * SA::SIZE;
        const float* bs = Bs0 + cur * SB::SIZE;
        // (after the last K tile this stores a zero tile nobody reads)
        sa.template store<MASKED, U>(As0 + (cur ^ 1) * SA::SIZE, klen - (t + 1) * BK);
        sb.template store<MASKED, U>(Bs0 + (cur ^ 1) * SB::SIZE, klen - (t + 1) * BK);
        sa.template load<MASKED, U>(klen - (t + 1 + NS) * BK); sb.template load<MASKED, U>(klen - (t + 1 + NS) * BK);
        __builtin_amdgcn_sched_barrier(0);     // hipcc otherwise sinks stores and loads down to the barrier
        float fa[2][TM][4], fb[2][TN][4];
#pragma unroll
        for (int i = 0; i < TM; ++i) fetch_group<BM, BK, SA::LK>(as, wm * WM + i * 32 + l31, 0, lhi, fa[0][i]);
#pragma unroll
        for (int j = 0; j < TN; ++j) fetch_group<BN, BK, SB::LK>(bs, wn * WN + j * 32 + l31, 0, lhi, fb[0][j]);
#pragma unroll
        for (int q = 0; q < G; ++q) {
            if (q + 1 < G) {
#pragma unroll
                for (int i = 0; i < TM; ++i)
                    fetch_group<BM, BK, SA::LK>(as, wm * WM + i * 32 + l31, q + 1, lhi, fa[(q + 1) & 1][i]);
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    fetch_group<BN, BK, SB::LK>(bs, wn * WN + j * 32 + l31, q + 1, lhi, fb[(q + 1) & 1][j]);
            }
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[q & 1][i][s], fb[q & 1][j][s],
                                                                         acc[i][j], 0, 0, 0);
        }
        __syncthreads();
        cur ^= 1;
    };
    auto k_group = [&](auto masked, long t) {            // NS consecutive tiles starting at a multiple of NS
        constexpr bool MASKED = decltype(masked)::value;  // (an unmasked group is complete by construction)
        k_tile(masked, std::integral_constant<int, 1 % NS>{}, t);
        if constexpr (NS > 1) { if (!MASKED || t + 1 < nt) k_tile(masked, std::integral_constant<int, 2 % NS>{}, t + 1); }
        if constexpr (NS > 2) { if (!MASKED || t + 2 < nt) k_tile(masked, std::integral_constant<int, 3 % NS>{}, t + 2); }
        if constexpr (NS > 3) { if (!MASKED || t + 3 < nt) k_tile(masked, std::integral_constant<int, 4 % NS>{}, t + 3); }
    };
    long t = 0;
    for (; (t + 2 * NS + 1) * BK <= klen; t += NS) k_group(std::false_type{}, t);   // every tile touched is full
    for (; t < nt; t += NS) k_group(std::true_type{}, t);

    // ---- split-K: publish the partial tile, the last arriver sums all slices in slice order ----------
    // (cdna_hip_programming.md, in-launch split-K hand-off: plain slab stores -> per-wave vmcnt(0) ->
    //  barrier -> one-lane agent-scope release -> ticket; reducer: one-lane agent-scope acquire ->
    //  barrier -> plain loads.)  Slab layout = the register image: float4 #(i,j,r4) of thread t.
    if (split > 1 && g.out_mode != 2) {
        float4* slab = reinterpret_cast<float4*>(g.slabs) +
                       ((size_t)(tile - g.t0) * split + slice) * (size_t)(BM * BN / 4);
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r4 = 0; r4 < 4; ++r4)
                {
                    const float4 v = make_float4(acc[i][j][4 * r4], acc[i][j][4 * r4 + 1], acc[i][j][4 * r4 + 2],
                                                 acc[i][j][4 * r4 + 3]);
                    float4* dst = slab + ((i * TN + j) * 4 + r4) * 256 + threadIdx.x;
#ifndef NEMO_GEMM_PLAIN_SLABS
                    // write-through (sc1) store: visible device-wide once vmcnt drains, no L2 write-back fence
                    const f32x4v vv = {v.x, v.y, v.z, v.w};
                    asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" :: "v"(dst), "v"(vv) : "memory");
#else
                    *dst = v;
#endif
                }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        int* flag = reinterpret_cast<int*>(smem);
        if (threadIdx.x == 0) {
#ifdef NEMO_GEMM_PLAIN_SLABS
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
            *flag = __hip_atomic_fetch_add(g.counters + tile, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        __syncthreads();
        if (*flag != split - 1) return;
        if (threadIdx.x == 0) {
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            __hip_atomic_store(g.counters + tile, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ready for the next launch
        }
        __syncthreads();
        const float4* base = reinterpret_cast<const float4*>(g.slabs) + (size_t)(tile - g.t0) * split * (size_t)(BM * BN / 4);
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r4 = 0; r4 < 4; ++r4) {
                    float4 sum = make_float4(0.f, 0.f, 0.f, 0.f);
                    for (int sl = 0; sl < split; ++sl) {          // fixed order; own slice re-read like the others
                        const float4 v = base[(size_t)sl * (BM * BN / 4) + ((i * TN + j) * 4 + r4) * 256 + threadIdx.x];
                        sum.x += v.x; sum.y += v.y; sum.z += v.z; sum.w += v.w;
                    }
                    acc[i][j][4 * r4] = sum.x; acc[i][j][4 * r4 + 1] = sum.y;
                    acc[i][j][4 * r4 + 2] = sum.z; acc[i][j][4 * r4 + 3] = sum.w;
                }
    }

    // Epilogue.  C/D layout of the 32x32 MFMA: col = lane & 31, row = (r&3) + 8*(r>>2) + 4*(lane>>5).
    const bool add_bias = g.bias != nullptr && (slice == 0 || g.out_mode != 2);
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const long n = n0 + wn * WN + j * 32 + l31;
            if (n >= g.N) continue;
            const float bv = add_bias ? g.bias[n] : 0.f;
            unsigned short tq[4] = {0, 0, 0, 0};
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const long m = m0 + wm * WM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
                if (m >= g.M) {
                    if (g.CbT) {
                        // a ragged last group of four rows: its valid members go out one by one
                        if ((r & 3) > 0 && m - (r & 3) < g.M)
                            for (int u = 0; u < (r & 3); ++u)
                                if (m - (r & 3) + u < g.M) g.CbT[n * g.ldcbt + m - (r & 3) + u] = tq[u];
                        // (odd M: column M of the transposed bf16 copy is the k-pad of the product that reads it)
                        if (m == g.M && (g.M & 1) && m < g.ldcbt) g.CbT[n * g.ldcbt + m] = 0;
                    }
                    continue;
                }
                float v = g.alpha * acc[i][j][r] + bv;
                if (g.act == 1) v = v > 0.f ? v : 0.f;
                else if (g.act == 2) v = v > 0.f ? v : 0.01f * v;
                if (g.mask_mode) {
                    const float mv = g.mask[m * g.ldmask + n];
                    if (g.mask_mode == 1) v = mv > 0.f ? v : 0.f;
                    else v = mv > 0.f ? v : 0.01f * v;
                }
                if (g.C) {
                    float* c = g.C + m * g.ldc + n;
                    if (g.out_mode == 0) *c = v;
                    else if (g.out_mode == 1) *c += v;
                    else atomicAdd(c, v);
                }
                // nemo_gemm_f32_b16out: the result as bf16 and as its bf16 transpose (fp32 arithmetic; the first layer of a
                // bf16-in-memory chain, whose own operands stay fp32)
                if (g.Cb || g.CbT) {
                    const unsigned short bits = __builtin_bit_cast(unsigned short, (__bf16)v);
                    if (g.Cb) g.Cb[m * g.ldcb + n] = bits;
                    if (g.CbT) {
                        // the lane's registers r = 4 q .. 4 q + 3 are four CONSECUTIVE rows m: one 8-byte store per group
                        tq[r & 3] = bits;
                        if ((r & 3) == 3) {
                            uint2 pk;
                            pk.x = (unsigned)tq[0] | ((unsigned)tq[1] << 16);
                            pk.y = (unsigned)tq[2] | ((unsigned)tq[3] << 16);
                            *reinterpret_cast<uint2*>(g.CbT + n * g.ldcbt + (m - 3)) = pk;
                        }
                    }
                }
            }
        }
}

#ifndef NS_BIG
#define NS_BIG 2
#endif
#ifndef NS_TN
#define NS_TN 2
#endif
template <int BM, int BN, int BK, bool TA, bool TB, bool VA, bool VB>
hipError_t launch_one(const GemmArgs& g, int blocks, hipStream_t s) {
    // register-pipeline depth: 2 K tiles in flight (measured per layout on MI355X, tools/bench_gemm.py; the
    // row-contiguous x row-contiguous layout ran deeper, 3, while its LDS image was k-major)
    constexpr int NS = BM == 64 ? ((TA && !TB) ? NS_TN : 2) : NS_BIG;
    using SA = Stager<BM, BK, !TA, VA, NS>;
    using SB = Stager<BN, BK, TB, VB, NS>;
    constexpr size_t lds = 2 * (SA::SIZE + SB::SIZE) * sizeof(float);
    static NemoAttrOnce attr_once;        // > 64 KB of LDS needs the opt-in once per instantiation
    if (attr_once.need()) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_f32_kernel<BM, BN, BK, TA, TB, VA, VB, NS>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL((gemm_f32_kernel<BM, BN, BK, TA, TB, VA, VB, NS>), dim3(blocks), dim3(256), lds, s, g);
    return hipSuccess;
}

template <int BM, int BN, int BK, bool VA, bool VB>
hipError_t launch(int ta, int tb, const GemmArgs& g, int blocks, hipStream_t s) {
    if (!ta && !tb) return launch_one<BM, BN, BK, false, false, VA, VB>(g, blocks, s);
    if (!ta && tb) return launch_one<BM, BN, BK, false, true, VA, VB>(g, blocks, s);
    if (ta && !tb) return launch_one<BM, BN, BK, true, false, VA, VB>(g, blocks, s);
    return launch_one<BM, BN, BK, true, true, VA, VB>(g, blocks, s);
}


inline bool aligned16(const void* p, long ld) { return (((uintptr_t)p) & 15) == 0 && (ld & 3) == 0; }

// Host cost model in microseconds, fitted to tools/bench_gemm.py --sweep on an MI355X (profiles/
// r01c_gemm_sweep.md): the busiest CU runs ceil(blocks / 256) blocks one after (or beside) the other at
// a fixed cost per K tile, and an in-launch combine costs a fixed hand-off plus the slab round trip.
constexpr int BK = 32;
constexpr long COUNTER_BYTES = 16384;       // 4096 tile tickets at the head of the workspace
constexpr int N_CU = 256;

struct Plan { int tile; int split; double cost; long t0; };

// LDS-DMA kernel, 64x64 tiles (tools/gemm_glds_dev calib, profiles/r02_gemm_glds.md): up to three blocks are resident
// per CU (48 KB of LDS each) and share its matrix pipes -- a K tile of every one of r co-resident blocks takes
// GLDS_US[r] microseconds (0.66 alone, 1.10 for a pair, 1.4 - 1.75 for three depending on what the operands are; 1.75 keeps the plan from leaning on three-way residency).
// `blocks` blocks of `it` K tiles each: whole rounds of 768, then the remainder at its own residency.
constexpr double GLDS_US[4] = {0.0, 0.66, 1.10, 1.75};
double glds_time(long blocks, long it) {
    double t = 0.0;
    while (blocks > 0) {
        const long g = blocks < 3 * N_CU ? blocks : 3 * N_CU;
        t += it * GLDS_US[(g + N_CU - 1) / N_CU];
        blocks -= g;
    }
    return t;
}

Plan plan_gemm(long M, long N, long K, bool can_split, long ws_bytes, bool atomic_mode, int forced_split,
               bool big_ok, bool vec, bool glds_ok) {
    Plan best{64, 1, 1e30, 0};
    if (glds_ok) {
        const long kiters = (K + BK - 1) / BK;
        const long tiles = ((M + 63) / 64) * ((N + 63) / 64);
        const int smax = forced_split > 0 ? forced_split : (can_split || atomic_mode ? 32 : 1);
        auto combine = [&](long slabs) { return 2.0 + (atomic_mode ? 1.0 : 2.0) * slabs * 16384.0 / 3.0e6; };
        for (int S = forced_split > 0 ? forced_split : 1; S <= smax; ++S) {
            if (S > 1 && forced_split <= 0 && kiters / S < 4) break;
            if (S > 1 && !atomic_mode && (tiles > COUNTER_BYTES / 4 || COUNTER_BYTES + tiles * S * 16384L > ws_bytes)) break;
            double c = glds_time(tiles * S, (kiters + S - 1) / S);
            if (S > 1) c += combine(tiles * S);
            if (c < best.cost) best = Plan{64, S, c, 0};
        }
        // whole tiles for every full set of co-resident blocks, only the remaining tiles cut along K
        if (can_split && !atomic_mode && forced_split <= 0 && tiles <= COUNTER_BYTES / 4)
            for (long per : {(long)N_CU, 2L * N_CU, 3L * N_CU}) {
                if (tiles <= per || tiles % per == 0) continue;
                const long t0 = (tiles / per) * per, tail = tiles - t0;
                for (int S = 2; S <= 16; ++S) {
                    if (kiters / S < 4 || COUNTER_BYTES + tail * S * 16384L > ws_bytes) break;
                    const double c = glds_time(t0, kiters) + glds_time(tail * S, (kiters + S - 1) / S) + combine(tail * S);
                    if (c < best.cost) best = Plan{64, S, c, t0};
                }
            }
        if (!big_ok) return best;
    }
    for (int tile : {128, 64}) {
        if (tile == 128 && !big_ok) continue;
        if (tile == 64 && glds_ok) continue;
        const int bk = BK;
        const long kiters = (K + bk - 1) / bk;
        const long tiles = ((M + tile - 1) / tile) * ((N + tile - 1) / tile);
        const double us_iter = tile == 128 ? 2.35 : 0.67 * bk / 32;   // one K tile of one block
        const int smax = forced_split > 0 ? forced_split : (can_split || atomic_mode ? 32 : 1);
        for (int S = forced_split > 0 ? forced_split : 1; S <= smax; ++S) {
            if (S > 1 && forced_split <= 0 && kiters / S < 4) break;
            if (S > 1 && !atomic_mode) {
                if (tiles > COUNTER_BYTES / 4) break;
                if (COUNTER_BYTES + tiles * S * (long)tile * tile * 4 > ws_bytes) break;
            }
            const long it = (kiters + S - 1) / S;
            const long blocks = tiles * S;
            const long per_cu = (blocks + N_CU - 1) / N_CU;
            double c = per_cu * it * us_iter;
            if (S > 1) c += 3.0 + (atomic_mode ? 1.0 : 2.0) * blocks * (double)tile * tile * 4 / 3.0e6;
            if (c < best.cost) best = Plan{tile, S, c, 0};
        }
        // whole tiles for every full round of 256, the remainder cut along K: 608 tiles cost what 768 cost
        // (profiles/r01d_gemm_quantisation.md); the remainder's slices fill one more, short, round instead
        if (tile == 64 && can_split && !atomic_mode && forced_split <= 0 && tiles > N_CU && tiles % N_CU) {
            const long t0 = (tiles / N_CU) * N_CU, tail = tiles - t0;
            for (int S = 2; S <= 16; ++S) {
                if (kiters / S < 4) break;
                if (COUNTER_BYTES + tail * S * (long)tile * tile * 4 > ws_bytes || tiles > COUNTER_BYTES / 4) break;
                const long it = (kiters + S - 1) / S;
                const double c = (t0 / N_CU) * kiters * us_iter + ((tail * S + N_CU - 1) / N_CU) * it * us_iter + 3.0 +
                                 2.0 * tail * S * (double)tile * tile * 4 / 3.0e6;
                if (c < best.cost) best = Plan{tile, S, c, t0};
            }
        }
    }
    return best;
}

}  // namespace

// Skinny problems (gemm_skinny.h): K split inside the block, no cross-block hand-off.  kind 0: 32 x 32 tiles on 4 waves
// (per-operand dword fall-backs for unaligned image-K operands), 1: 32 x 64 tiles on 8 waves, 2: 64 x 64 tiles on 8 waves
// (the 1000 x 1000 parameter gradients), 3: 32 x 32 tiles on 8 waves with K also cut across blocks (Args::split; the
// blend-shape adjoint of a one-instance shard); 1 - 3 need 16-byte aligned image-K operands.
static hipError_t skinny_launch(bool akc, bool bkc, bool va, bool vb, int kind, const GemmArgs& g, hipStream_t s) {
    // alignment only matters for image-K operands
    const bool av = va || !akc, bv = vb || !bkc;
#define SK(AKC, BKC)                                                                                    \
    do {                                                                                                \
        if (kind == 3) e = skinny::launch<AKC, BKC, true, true, 1, 1, 8, 4>(g, s);                      \
        else if (kind == 2) e = skinny::launch<AKC, BKC, true, true, 2, 2, 8, 3>(g, s);                 \
        else if (kind == 1) e = skinny::launch<AKC, BKC, true, true, 1, 2, 8, 4>(g, s);                 \
        else if (av && bv) e = skinny::launch<AKC, BKC, true, true, 1, 1, 4, 4>(g, s);                  \
        else if (av) e = skinny::launch<AKC, BKC, true, !(BKC), 1, 1, 4, 4>(g, s);                      \
        else if (bv) e = skinny::launch<AKC, BKC, !(AKC), true, 1, 1, 4, 4>(g, s);                      \
        else e = skinny::launch<AKC, BKC, !(AKC), !(BKC), 1, 1, 4, 4>(g, s);                            \
    } while (0)
    hipError_t e = hipSuccess;
    if (kind == 4) return skinny::launch<false, true, true, true, 1, 7, 4, 2>(g, s);        // (TT adjoint only)
    if (akc && bkc) SK(true, true);
    else if (akc) SK(true, false);
    else if (bkc) SK(false, true);
    else SK(false, false);
    return e;
#undef SK
}

// Which problems go to the skinny kernel (-1: none), from tools/gemm_skinny_dev sweeps at 300 / 600 / 1200 / 2400 rows
// (profiles/r02_gemm_skinny.md).  Image-M operands (dword loads, two full cache lines per instruction) stream well at any
// size; image-K operands cost the texture path 32 lines per dwordx4 instruction, so the layouts that have them hand over
// to the LDS-staged kernel (whose LDS-DMA pieces are whole lines) once its 64 x 64 grid fills the chip.
static int skinny_kind(bool akc, bool bkc, bool aligned, bool can_split, long M, long N, long K) {
    const long t64 = ((M + 63) / 64) * ((N + 63) / 64), t3264 = ((M + 31) / 32) * ((N + 63) / 64);
    bool use;
    if (!akc && !bkc) {                     // parameter gradients dW = dY^T X
        // (at small K these run beside the input-gradient chain on a second stream: the 64 x 64 configuration's 128 KiB
        //  reduction buffer would keep the other stream's blocks off the CU)
        // (tools/debug/dw_bigk.py, 1000 x 1000, us per launch: K = 2401: 61 against 65 for the LDS-staged kernel's K split;
        //  12 001: 256 / 264; 32 768: 759 / 734; 65 536: 1443 / 1399; 131 072: 2671 / 2634; 262 145 (C4): 5364 / 4924 -- every
        //  workgroup streams its own two 64-column strips of the operands, 16 x their size in all, and at C4 that is
        //  33 GB per launch at 6.5 TB/s: HBM-bound; the K-split plan's workgroups of one slice share theirs in L2)
        if (t64 >= 200 && t64 <= 256 && K >= 1024 && K < 32768) return 2;
        // (the 1000 x 1000 gradients of a two-instance shard, K = 600: 0.642 -> 0.635 ms per step on the 32 x 32
        //  configuration; at K = 300 the LDS-DMA kernel is 3 us ahead)
        use = (t64 <= 192 || (t64 <= 256 && K >= 512)) && K <= 4096;
    } else if (akc && !bkc) use = ((M <= 1536 && t64 <= 320) || t64 <= 192) && K <= 4096;   // input gradients dX = dY W
    else if (akc && bkc) use = M <= 1024 && t64 <= 192 && K <= 4096;         // forward Y = X W^T
    else {
        // dPF = dVP P^T of a one-instance shard (K = 20 670 over 70 tiles): 58 us with 16 K slices across blocks against
        // 68 us for the LDS-staged kernel's best plan; from ~600 samples on that kernel wins (profiles/r02_gemm_skinny.md)
        // two-instance shards (400 - 1000 samples; round 3): ONE column tile of 224 -- 8 % padding instead of the 24 % of
        // four 64-wide tiles -- on 4 waves, K cut into 256 / row-tiles slices: 72 us at 600 samples against 91 (kind 3)
        // and 84 (LDS-staged kernel, 6 slices); at 300 samples all three meet at 53 - 58 us, from 1200 on the LDS-staged
        // kernel is level (tools/gemm_skinny_dev adj, profiles/r03_experiments.md)
        if (K > 4096 && aligned && can_split && N > 192 && N <= 224 && M > 384 && M <= 1000) return 4;
        if (K > 4096) return aligned && can_split && t64 <= 24 ? 3 : -1;
        use = t64 <= 192;
    }
    if (!use) return -1;
    return aligned && K >= 512 && t3264 >= 128 && t3264 <= 256 ? 1 : 0;
}

// bf16: 0 = fp32 arithmetic; 1 = fp32 operands in memory, rounded to bf16 on their way into the matrix cores; 2 = operands
// ALREADY bf16 in memory (NT only: A (M x K) and B (N x K) k-contiguous; A / B / lda / ldb / K arrive in units of bf16
// PAIRS, i.e. as the fp32-typed view of the same bytes), optional bf16 copies of the result (Cb, CbT)
// colsum[band][n] = sum of rows [32 band, 32 band + 32) of C (the scratch layout of the GEMM epilogues' per-band column sums)
__global__ __launch_bounds__(64) void band_colsum_kernel(const float* __restrict__ C, long M, long N, long ldc, float* __restrict__ colsum,
                                                         long ldcs) {
    const long n = (long)blockIdx.x * 64 + threadIdx.x, m0 = (long)blockIdx.y * 32;
    if (n >= N) return;
    float s = 0.f;
    for (long m = m0; m < min(M, m0 + 32); ++m) s += C[m * ldc + n];
    colsum[(long)blockIdx.y * ldcs + n] = s;
}

static int32_t gemm_impl(int bf16, int32_t transA, int32_t transB, int64_t M, int64_t N, int64_t K,
                         const float* A, int64_t lda, const float* B, int64_t ldb,
                         float* C, int64_t ldc, const float* bias, int32_t act,
                         const float* mask, int64_t ldmask, int32_t mask_mode, float alpha,
                         int32_t out_mode, int32_t split_k, void* ws, int64_t ws_bytes, void* stream,
                         unsigned short* Cb = nullptr, long ldcb = 0, unsigned short* CbT = nullptr, long ldcbt = 0,
                         const unsigned short* mask16 = nullptr, long ldmask16 = 0, float* colsum = nullptr, long ldcs = 0) {
    // bf16 == 0 with Cb / CbT (nemo_gemm_f32_b16out): fp32 product whose result is (also) stored as bf16 -- register-staged
    // kernel only (its epilogue writes the copies; the operands of that call are the unaligned ones anyway)
    const bool b16out = bf16 == 0 && (Cb || CbT);
    if (M < 0 || N < 0 || K < 0 || (!C && !((bf16 == 2 || b16out) && (Cb || CbT) && out_mode == 0))) return NEMO_EINVAL;
    if (b16out && (out_mode != 0 || (Cb && ldcb < N) || (CbT && (ldcbt < M || (ldcbt & 3) || (((uintptr_t)CbT) & 7))))) return NEMO_EINVAL;
    // (column sums of the result per 32-row band: the bf16-in-memory kernel's epilogue since round 3, the fp32 LDS-DMA kernel's
    //  since round 5 -- nemo_gemm_f32_colsum; other plans compute them from C in a launch of their own, band_colsum_kernel)
    if (colsum && (bf16 == 1 || out_mode == 2 || (bf16 == 0 && (out_mode != 0 || !C)) || ldcs < N)) return NEMO_EINVAL;
    const bool cs32 = colsum != nullptr && bf16 == 0;
    if (M == 0 || N == 0) return NEMO_OK;
    if (K > 0 && (!A || !B)) return NEMO_EINVAL;
    if (act < 0 || act > 2 || mask_mode < 0 || mask_mode > 2 || out_mode < 0 || out_mode > 2)
        return NEMO_EINVAL;
    if (mask_mode && !mask && !mask16) return NEMO_EINVAL;
    if (split_k < 0 || ws_bytes < 0 || (ws_bytes > 0 && !ws)) return NEMO_EINVAL;
    // atomically accumulated slices are only linear: no activation / mask
    if (out_mode == 2 && (act || mask_mode)) return NEMO_EINVAL;
    const bool can_split = ws != nullptr && ws_bytes > COUNTER_BYTES && (((uintptr_t)ws) & 15) == 0;
    if (split_k > 1 && out_mode != 2 && !can_split) return NEMO_EINVAL;

    int force_tile = 0;
    if (const char* f = getenv("NEMO_GEMM_TILE")) force_tile = atoi(f);          // test / tuning aid (tests/test_gpu_ops.py, tools/bench_gemm.py)
    // K-chunk starts are multiples of 32, so operand alignment only depends on the base and the ld; each
    // operand independently takes the dwordx4 or the dword staging path (64x64 tiles; the 128x128 tile
    // needs both aligned).  Dword staging serves the few operands whose rows are not 16-byte aligned,
    // e.g. nn.Linear weights with in_features = 105 or a view that starts 3 floats into a row.
    const bool va = aligned16(A, lda), vb = aligned16(B, ldb);
    const bool vec = va && vb;
    // (128x128 tiles pay off only with both operands k-contiguous: the k-major LDS image is fetched with
    //  four ds_read_b32 per operand group instead of one ds_read_b128 and runs at ~half the rate there)
    unsigned a_bytes = 0, b_bytes = 0;
    const bool glds_ok = vec && force_tile != 128 && !b16out &&
                         glds::extents(transA, transB, M, N, K, lda, ldb, &a_bytes, &b_bytes);
    Plan pl = plan_gemm(M, N, K, can_split, ws_bytes, out_mode == 2, split_k,
                        vec && !transA && transB && !(bf16 && glds_ok), vec, glds_ok);
    if (vec && (force_tile == 64 || force_tile == 128)) {
        pl.tile = force_tile;
        pl.t0 = 0;
        if (split_k > 0) pl.split = split_k;
    }
    // The blend-shape adjoint (dPF = dVP P^T: M = samples, N = 207, K = 3 NV = 20 670, both operands "transposed") from 256
    // samples on: ONE 64 x 208 column tile per workgroup with mixed MFMA shapes (gemm_adj.h: columns [0, 192) on
    // v_mfma_f32_32x32x2_f32, the 16-column remainder on 16x16x4) -- dVP^T is streamed once instead of once per 64-column
    // tile and 0.5 % instead of 24 % of the MFMAs multiply padding.  tools/gemm_glds_dev adj <M> (us per launch, best K
    // split each): M = 300: 49 against 58 (skinny 32 x 32 x 16 slices); 512: 56 against 72 (skinny 32 x 224); 600: 66 against 72;
    // 1200: 101 against 130 (64 x 64 plan) / 128; 2400: 172 against 227; 3808: 270 against 305 (round 2's all-16x16x4 wide
    // tile) / 351; 8192: 526 against 570 / 688.  K slices so that one workgroup per CU (up to 8 row tiles) or two (beyond) are
    // resident -- a count that overshoots the 256 / 512 slots by a few workgroups costs a whole extra round -- and every slice
    // keeps >= 8 K tiles; beyond 16 slices they are summed in two levels (gemm_adj.h).
    if (glds_ok && !bf16 && transA && transB && N > 128 && N <= 208 && M >= 256 && K >= 2048 && split_k == 0 &&
        out_mode != 2 && force_tile == 0 && can_split && !bias && !act && !mask_mode && !colsum) {
        const long tiles_m = (M + 63) / 64;
        int S = (int)((tiles_m <= 8 ? 256 : 512) / tiles_m);
        if (S < 1) S = 1;
        while (S > 1 && ((K + 31) / 32 / S < 8 || COUNTER_BYTES + glds::adj_slab_floats(tiles_m, S) * 4 > ws_bytes ||
                         glds::adj_counter_ints(tiles_m, S) > COUNTER_BYTES / 4)) --S;
        if (glds::adj_counter_ints(tiles_m, S) <= COUNTER_BYTES / 4) {
            GemmArgs g;
            g.A = A; g.B = B; g.C = C; g.bias = nullptr; g.mask = nullptr;
            g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldb = ldb; g.ldc = ldc; g.ldmask = 0;
            g.act = 0; g.mask_mode = 0; g.out_mode = out_mode; g.alpha = alpha;
            g.counters = reinterpret_cast<int*>(ws);
            g.slabs = reinterpret_cast<float*>(reinterpret_cast<char*>(ws) + COUNTER_BYTES);
            long kc = (K + S - 1) / S;
            kc = ((kc + 31) / 32) * 32;
            g.k_chunk = kc;
            g.split = (int)((K + kc - 1) / kc);
            g.tiles_m = (int)tiles_m; g.tiles_n = 1; g.n_tiles = (int)tiles_m; g.t0 = 0;
            g.a_bytes = a_bytes; g.b_bytes = b_bytes; g.xcd_order = 0;
            static const bool debug_wide = getenv("NEMO_GEMM_DEBUG") != nullptr;
            if (debug_wide)
                fprintf(stderr, "nemo_gemm_f32 ta=1 tb=1 M=%ld N=%ld K=%ld -> 64x208 mixed-shape tile, %d K slices\n", (long)M, (long)N, (long)K, g.split);
            hipError_t e = glds::launch_adj(g, (hipStream_t)stream);
            if (e != hipSuccess) return (int32_t)e;
            NEMO_LAUNCH_CHECK();
            return NEMO_OK;
        }
    }
    // The same tile for the bf16-in-memory chain: dPF (+)= dVP P^T with both operands bf16 and k-contiguous (gemm_adj.h B16).
    // K counts bf16 pairs here.
    if (glds_ok && bf16 == 2 && !transA && transB && N > 128 && N <= 208 && M >= 256 && K >= 1024 && split_k == 0 && C &&
        out_mode != 2 && force_tile == 0 && can_split && !bias && !act && !mask_mode && !mask16 && !Cb && !CbT && !colsum) {
        // (from 1024 rows on: the 128 x 208 tile on eight waves, one workgroup per CU -- as nemo_gemm_f16x2mem_adj; NEMO_ADJ128=0: never)
        static const bool no128b = getenv("NEMO_ADJ128") != nullptr && atoi(getenv("NEMO_ADJ128")) == 0;
        const bool t128 = M >= 1024 && !no128b;
        const int bm = t128 ? 128 : 64;
        const long tiles_m = (M + bm - 1) / bm;
        int S = (int)(((t128 || tiles_m <= 8) ? 256 : 512) / tiles_m);
        if (S < 1) S = 1;
        while (S > 1 && ((K + 31) / 32 / S < 8 || COUNTER_BYTES + glds::adj_slab_floats(tiles_m, S, bm) * 4 > ws_bytes ||
                         glds::adj_counter_ints(tiles_m, S) > COUNTER_BYTES / 4)) --S;
        if (glds::adj_counter_ints(tiles_m, S) <= COUNTER_BYTES / 4 &&
            COUNTER_BYTES + glds::adj_slab_floats(tiles_m, S, bm) * 4 <= ws_bytes) {
            GemmArgs g;
            g.A = A; g.B = B; g.C = C; g.bias = nullptr; g.mask = nullptr;
            g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldb = ldb; g.ldc = ldc; g.ldmask = 0;
            g.act = 0; g.mask_mode = 0; g.out_mode = out_mode; g.alpha = alpha;
            g.counters = reinterpret_cast<int*>(ws);
            g.slabs = reinterpret_cast<float*>(reinterpret_cast<char*>(ws) + COUNTER_BYTES);
            long kc = (K + S - 1) / S;
            kc = ((kc + 31) / 32) * 32;
            g.k_chunk = kc;
            g.split = (int)((K + kc - 1) / kc);
            g.tiles_m = (int)tiles_m; g.tiles_n = 1; g.n_tiles = (int)tiles_m; g.t0 = 0;
            g.a_bytes = a_bytes; g.b_bytes = b_bytes; g.xcd_order = 0;
            static const bool debug_b16 = getenv("NEMO_GEMM_DEBUG") != nullptr;
            if (debug_b16)
                fprintf(stderr, "nemo_gemm_bf16mem M=%ld N=%ld K=%ld (bf16 pairs) -> 64x208 mixed-shape tile, %d K slices\n", (long)M, (long)N, (long)K, g.split);
            hipError_t e = t128 ? glds::launch_adj128_b16(g, (hipStream_t)stream) : glds::launch_adj(g, (hipStream_t)stream, true);
            if (e != hipSuccess) return (int32_t)e;
            NEMO_LAUNCH_CHECK();
            return NEMO_OK;
        }
    }
    // Skinny problems -- the 64 x 64 grid could not fill the chip without cutting K across blocks (one-instance shards,
    // mini-batches of a few hundred samples): the intra-block K split of gemm_skinny.h, 19 -> 13 us for 300 x 1000 x 1000
    // and ~2x on the small layers (profiles/r02_gemm_skinny.md).  Very long K (the blend-shape adjoint) stays with the
    // LDS-staged kernel: a wave's K slice would be thousands of steps.
    const int sk_kind = !bf16 && !b16out && !colsum && split_k == 0 && force_tile == 0 && K >= 1
                            ? skinny_kind(!transA, transB != 0, (va || transA) && (vb || !transB),
                                          can_split && out_mode != 2 && ws_bytes >= COUNTER_BYTES + (8L << 20), M, N, K) : -1;
    if (sk_kind >= 0 && (glds_ok || glds::extents(transA, transB, M, N, K, lda, ldb, &a_bytes, &b_bytes))) {
        GemmArgs g;
        g.A = A; g.B = B; g.C = C; g.bias = bias; g.mask = mask;
        g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldb = ldb; g.ldc = ldc; g.ldmask = ldmask;
        g.act = act; g.mask_mode = mask_mode; g.out_mode = out_mode; g.alpha = alpha;
        g.counters = nullptr; g.slabs = nullptr; g.k_chunk = (K + 7) / 8 * 8; g.split = 1; g.n_tiles = 0; g.t0 = 0; g.xcd_order = 0;
        g.a_bytes = a_bytes; g.b_bytes = b_bytes;
        if (sk_kind == 4) {         // 256 / row-tiles K slices of a 32 x 224 tile: <= 256 slabs of 28 KiB = 7 MiB
            const long tiles = (M + 31) / 32;
            long kc = (K + 256 / tiles - 1) / (256 / tiles);
            kc = (kc + 7) / 8 * 8;
            g.k_chunk = kc;
            g.split = (int)((K + kc - 1) / kc);
            g.counters = reinterpret_cast<int*>(ws);
            g.slabs = reinterpret_cast<float*>(reinterpret_cast<char*>(ws) + COUNTER_BYTES);
        } else if (sk_kind == 3) {  // 16 K slices: <= 96 tiles of 32 x 32 x 16 slabs of 4 KiB = 6 MiB of the workspace
            long kc = (K + 15) / 16;
            kc = (kc + 7) / 8 * 8;
            g.k_chunk = kc;
            g.split = (int)((K + kc - 1) / kc);
            g.counters = reinterpret_cast<int*>(ws);
            g.slabs = reinterpret_cast<float*>(reinterpret_cast<char*>(ws) + COUNTER_BYTES);
        }
        static const bool debug_skinny = getenv("NEMO_GEMM_DEBUG") != nullptr;
        if (debug_skinny)
            fprintf(stderr, "nemo_gemm_f32 ta=%d tb=%d M=%ld N=%ld K=%ld out=%d -> skinny %s\n", transA, transB, (long)M, (long)N,
                    (long)K, out_mode, sk_kind == 4 ? "32x224 w4 x K slices" : sk_kind == 3 ? "32x32 w8 x 16 K slices" : sk_kind == 2 ? "64x64 w8" : sk_kind == 1 ? "32x64 w8" : "32x32 w4");
        const hipError_t e = skinny_launch(!transA, transB != 0, va, vb, sk_kind, g, (hipStream_t)stream);
        if (e != hipSuccess) return (int32_t)e;
        NEMO_LAUNCH_CHECK();
        return NEMO_OK;
    }
    const int tile = pl.tile;
    if (cs32 && !(tile == 64 && glds_ok)) {                 // (unaligned operands, the 128 x 128 tile: sums from C afterwards)
        const int32_t rc = gemm_impl(bf16, transA, transB, M, N, K, A, lda, B, ldb, C, ldc, bias, act, mask, ldmask, mask_mode, alpha,
                                     out_mode, split_k, ws, ws_bytes, stream);
        if (rc != NEMO_OK) return rc;
        hipLaunchKernelGGL(band_colsum_kernel, dim3(nemo_cdiv(N, 64), (unsigned)(2 * ((M + 63) / 64))), dim3(64), 0, (hipStream_t)stream,
                           C, (long)M, (long)N, (long)ldc, colsum, (long)ldcs);
        NEMO_LAUNCH_CHECK();
        return NEMO_OK;
    }
    static const bool debug_plans = getenv("NEMO_GEMM_DEBUG") != nullptr;          // tuning aid: the plan of every call
    if (debug_plans)
        fprintf(stderr, "nemo_gemm_f32 ta=%d tb=%d M=%ld N=%ld K=%ld out=%d -> %s tile %d split %d t0 %ld (model %.1f us)\n",
                transA, transB, (long)M, (long)N, (long)K, out_mode, glds_ok && tile == 64 ? "glds" : "v1", tile, pl.split,
                pl.t0, pl.cost);

    GemmArgs g;
    g.A = A; g.B = B; g.C = C; g.bias = bias; g.mask = mask;
    g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldb = ldb; g.ldc = ldc; g.ldmask = ldmask;
    g.act = act; g.mask_mode = mask_mode; g.out_mode = out_mode; g.alpha = alpha;
    g.counters = reinterpret_cast<int*>(ws);
    g.slabs = reinterpret_cast<float*>(reinterpret_cast<char*>(ws) + COUNTER_BYTES);
    long kc = (K + pl.split - 1) / pl.split;
    const int bk = BK;
    kc = ((kc + bk - 1) / bk) * bk;
    if (kc == 0) kc = bk;
    g.k_chunk = kc;
    int nz = (int)((K + kc - 1) / kc);
    if (nz < 1) nz = 1;
    g.split = nz;
    g.tiles_m = (int)((M + tile - 1) / tile); g.tiles_n = (int)((N + tile - 1) / tile);
    g.n_tiles = g.tiles_m * g.tiles_n;
    g.t0 = (nz > 1 && out_mode != 2 && pl.t0 > 0 && pl.t0 < g.n_tiles) ? (int)pl.t0 : 0;
    if (nz > 1 && out_mode != 2 &&
        (g.n_tiles > COUNTER_BYTES / 4 || COUNTER_BYTES + (long)(g.n_tiles - g.t0) * nz * tile * tile * 4 > ws_bytes))
        return NEMO_EINVAL;
    const long blocks = g.t0 + (long)(g.n_tiles - g.t0) * nz;
    if (blocks > 0x7fffffffL) return NEMO_EINVAL;

    hipStream_t s = (hipStream_t)stream;
    hipError_t e;
    // Second-generation kernel for the common case (64x64 tiles, both operands 16-byte aligned, extents that fit a
    // 32-bit buffer descriptor): operand tiles go global -> LDS by LDS-DMA, three stages, the pieces of the tile being
    // requested issued between the MFMAs of the tile being multiplied.  Same tiles, slices, slabs and epilogue as the
    // register-staged kernel below, which keeps the unaligned operands, the 128x128 tile and very large operands.
    g.xcd_order = 0;
    g.Cb = Cb; g.ldcb = ldcb; g.CbT = CbT; g.ldcbt = ldcbt; g.mask16 = mask16; g.ldmask16 = ldmask16;
    g.colsum = colsum; g.ldcs = ldcs;
    // (bf16-in-memory products on 128 x 128 / 128 x 64 tiles were measured SLOWER than the 64 x 64 tile at every shape of the
    //  step -- 12 000 x 1000 x 1000: 98 us = 244 TFLOP/s against 71 us = 337; profiles/r03_experiments.md section 9)
    if (bf16 == 2 && !(tile == 64 && glds_ok && !transA && transB)) return NEMO_EINVAL;
    if (tile == 64 && glds_ok) {
        g.a_bytes = a_bytes; g.b_bytes = b_bytes;
        long nblocks = blocks;
        if (nz == 1 && g.n_tiles >= 2048) {          // XCD-aware tile order for the large whole-tile launches
            g.xcd_order = 1;
            nblocks = 8L * ((g.tiles_m + 7) / 8) * g.tiles_n;
        }
        const bool akc = !transA, bkc = transB != 0;
        if (bf16 == 2) e = glds::launch<64, 64, 32, 32, 32, true, true, 3, true, 2>(g, (int)nblocks, s);
        else if (bf16) {
            if (akc && bkc) e = glds::launch<64, 64, 32, 32, 32, true, true, 3, true, true>(g, (int)nblocks, s);
            else if (akc) e = glds::launch<64, 64, 32, 32, 32, true, false, 3, true, true>(g, (int)nblocks, s);
            else if (bkc) e = glds::launch<64, 64, 32, 32, 32, false, true, 3, true, true>(g, (int)nblocks, s);
            else e = glds::launch<64, 64, 32, 32, 32, false, false, 3, true, true>(g, (int)nblocks, s);
        } else
        if (akc && bkc) e = glds::launch<64, 64, 32, 32, 32, true, true, 3, true>(g, (int)nblocks, s);
        else if (akc) e = glds::launch<64, 64, 32, 32, 32, true, false, 3, true>(g, (int)nblocks, s);
        else if (bkc) e = glds::launch<64, 64, 32, 32, 32, false, true, 3, true>(g, (int)nblocks, s);
        else e = glds::launch<64, 64, 32, 32, 32, false, false, 3, true>(g, (int)nblocks, s);
        if (e != hipSuccess) return (int32_t)e;
        NEMO_LAUNCH_CHECK();
        return NEMO_OK;
    }
    if (tile == 128) e = launch<128, 128, BK, true, true>(transA, transB, g, (int)blocks, s);
    else if (va && vb) e = launch<64, 64, BK, true, true>(transA, transB, g, (int)blocks, s);
    else if (va) e = launch<64, 64, BK, true, false>(transA, transB, g, (int)blocks, s);
    else if (vb) e = launch<64, 64, BK, false, true>(transA, transB, g, (int)blocks, s);
    else e = launch<64, 64, BK, false, false>(transA, transB, g, (int)blocks, s);
    if (e != hipSuccess) return (int32_t)e;
    NEMO_LAUNCH_CHECK();
    return NEMO_OK;
}

extern "C" int32_t nemo_gemm_f32(int32_t transA, int32_t transB, int64_t M, int64_t N, int64_t K,
                                 const float* A, int64_t lda, const float* B, int64_t ldb,
                                 float* C, int64_t ldc, const float* bias, int32_t act,
                                 const float* mask, int64_t ldmask, int32_t mask_mode, float alpha,
                                 int32_t out_mode, int32_t split_k, void* ws, int64_t ws_bytes, void* stream) {
    return gemm_impl(false, transA, transB, M, N, K, A, lda, B, ldb, C, ldc, bias, act, mask, ldmask, mask_mode, alpha,
                     out_mode, split_k, ws, ws_bytes, stream);
}

extern "C" int32_t nemo_gemm_f32_colsum(int32_t transA, int32_t transB, int64_t M, int64_t N, int64_t K,
                                        const float* A, int64_t lda, const float* B, int64_t ldb,
                                        float* C, int64_t ldc, const float* bias, int32_t act,
                                        const float* mask, int64_t ldmask, int32_t mask_mode, float alpha,
                                        float* colsum, int64_t ldcs, void* ws, int64_t ws_bytes, void* stream) {
    if (!colsum) return NEMO_EINVAL;
    return gemm_impl(false, transA, transB, M, N, K, A, lda, B, ldb, C, ldc, bias, act, mask, ldmask, mask_mode, alpha,
                     0, 0, ws, ws_bytes, stream, nullptr, 0, nullptr, 0, nullptr, 0, colsum, ldcs);
}

extern "C" int32_t nemo_gemm_bf16(int32_t transA, int32_t transB, int64_t M, int64_t N, int64_t K,
                                  const float* A, int64_t lda, const float* B, int64_t ldb,
                                  float* C, int64_t ldc, const float* bias, int32_t act,
                                  const float* mask, int64_t ldmask, int32_t mask_mode, float alpha,
                                  int32_t out_mode, int32_t split_k, void* ws, int64_t ws_bytes, void* stream) {
    return gemm_impl(true, transA, transB, M, N, K, A, lda, B, ldb, C, ldc, bias, act, mask, ldmask, mask_mode, alpha,
                     out_mode, split_k, ws, ws_bytes, stream);
}

// ---- bf16 operands IN MEMORY (round 3, BASELINE configs[2]) ----------------------------------------------------
// nemo_gemm_bf16 rounds fp32 operands to bf16 between LDS and the matrix cores: the LDS-DMA stream of fp32 tiles bounds
// it at ~190 TFLOP/s.  Here the operands are bf16 already (half the bytes through the DMA and LDS, no conversion
// VALU work): one layout -- both operands k-contiguous, C = A B^T -- which every product of the MLP chain takes once
// the caller keeps a bf16 copy and a TRANSPOSED bf16 copy of each activation / weight (the kernel's epilogue writes
// both for its own result: Cb, CbT; nemo_cast_bf16 makes them for everything else).  Same values enter the MFMAs as in
// nemo_gemm_bf16 (round-to-nearest-even of the same fp32 numbers), same fp32 accumulation.
extern "C" int32_t nemo_gemm_bf16mem(int64_t M, int64_t N, int64_t K, const uint16_t* A, int64_t lda, const uint16_t* B,
                                     int64_t ldb, float* C, int64_t ldc, const float* bias, int32_t act, const float* mask,
                                     int64_t ldmask, int32_t mask_mode, float alpha, int32_t out_mode, uint16_t* Cb,
                                     int64_t ldcb, uint16_t* CbT, int64_t ldcbt, float* colsum, int64_t ldcs, void* ws,
                                     int64_t ws_bytes, void* stream) {
    if (M < 0 || N < 0 || K < 0 || (K & 1) || (lda & 7) || (ldb & 7) || lda < K || ldb < K) return NEMO_EINVAL;
    if (!C && !(Cb || CbT)) return NEMO_EINVAL;
    if ((((uintptr_t)A) | ((uintptr_t)B)) & 15) return NEMO_EINVAL;
    if (Cb && ldcb < N) return NEMO_EINVAL;
    if (CbT && (ldcbt < M || (ldcbt & 3) || (((uintptr_t)CbT) & 7))) return NEMO_EINVAL;
    if (M == 0 || N == 0) return NEMO_OK;
    if (K == 0) return NEMO_EINVAL;
    // mask_mode 1 / 2 with a FLOAT mask; 17 / 18: the same tests on a BF16 mask (`mask` then points at uint16_t data and
    // ldmask counts bf16 elements) -- e.g. the bf16 copy of a ReLU output, whose sign and zeros survive the rounding
    const bool m16 = mask_mode >= 16;
    // Round 5: the large products of the chain go to the large-tile kernel (gemm_b16x.h: 8 MFMA waves on a 192 x 256 /
    // 128 x 256 tile + 4 loader waves).  tools/gemm_b16x_dev time, us per launch against the 64 x 64 kernel below
    // (12 001 x 1000 x 1000): forward with both bf16 copies 40.7 / 84.7, dX with mask, copies and column sums 45.8 / 110.7,
    // head dX (K = 152) 24.4 / 66.5; from 4801 rows on the 128 x 256 tile (26.0 / 34.4, dX 30.1 / 49.3), level at 2401 rows.
    // The parameter gradients (K = samples) only from 24 576 samples on (65 537: 235 / 363; 12 001: 63 / 67, 8193: 55 / 46):
    // every workgroup streams its own K slice there and a CU pulls ~40 GB/s.
    static const bool no_b16x = getenv("NEMO_B16X") != nullptr && atoi(getenv("NEMO_B16X")) == 0;
    // (N >= 768: the 512-wide VPoser layers run beside the key-point branch of the step; a kernel that takes whole CUs -- 120+ KiB
    //  of LDS, 768 threads -- got 100 us there for what the 64 x 64 kernel does in 65)
    static const long deep_k = getenv("NEMO_B16X_DEEP_K") ? atol(getenv("NEMO_B16X_DEEP_K")) : 24576;
    const bool wide = M >= 3072 && N >= 768 && K >= 64;
    const bool deep = K >= deep_k && M >= 512 && N >= 512 && M * N <= 2048L * 2048 && !Cb && !CbT && !colsum;
    if (!no_b16x && (wide || deep) && (mask_mode == 0 || (mask_mode == 17 && mask)) && out_mode != 2 &&
        (!Cb || ((ldcb & 7) == 0 && ldcb >= (N + 7) / 8 * 8 && (((uintptr_t)Cb) & 15) == 0)) &&
        (!CbT || ((ldcbt & 7) == 0 && ldcbt >= (M + 7) / 8 * 8 && (((uintptr_t)CbT) & 15) == 0)) &&
        (mask_mode == 0 || ((ldmask & 7) == 0 && ldmask >= N && (((uintptr_t)mask) & 15) == 0)) && (!colsum || ldcs >= N) &&
        !getenv("NEMO_GEMM_TILE")) {
        b16x::Args g{};
        g.A = A; g.B = B; g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldb = ldb; g.C = C; g.ldc = ldc; g.out_mode = out_mode;
        g.bias = bias; g.act = act; g.alpha = alpha;
        g.mask16 = mask_mode ? reinterpret_cast<const unsigned short*>(mask) : nullptr; g.ldmask16 = ldmask; g.mask_mode = mask_mode ? 1 : 0;
        g.Cb = Cb; g.ldcb = ldcb; g.CbT = CbT; g.ldcbt = ldcbt; g.colsum = colsum; g.ldcs = ldcs;
        g.counters = reinterpret_cast<int*>(ws);
        g.slabs = reinterpret_cast<float*>(reinterpret_cast<char*>(ws) + COUNTER_BYTES);
        const long t128 = ((M + 127) / 128) * ((N + 255) / 256), t192 = ((M + 191) / 192) * ((N + 255) / 256);
        // (whole-line ring stages, BK = 64: 38.5 -> 35.6 us for the 12 001-row forward with both copies, 691 -> 762 TFLOP/s in the
        //  evaluation form, 128 x 256: 53.2 -> 45.2 us; NEMO_B16X_BK=32: the 64-byte-row form)
        static const bool bk32 = getenv("NEMO_B16X_BK") != nullptr && atoi(getenv("NEMO_B16X_BK")) == 32;
        int cfg = bk32 ? 3 : 6, split = 1;                  // 192 x 256 + loader waves
        if (wide && t128 <= 256) cfg = bk32 ? 5 : 8;        // 128 x 256 + loader waves while its grid is one round of the chip
        if (!wide) {                                        // K slices: ~one workgroup per CU, >= 2048 k each
            split = (int)(256 / t192);
            while (split > 1 && (K / split < 2048 || COUNTER_BYTES + t192 * split * 192L * 256 * 4 > ws_bytes)) --split;
            if (!ws || (((uintptr_t)ws) & 15) || t192 > COUNTER_BYTES / 4) split = 1;
        }
        if (b16x::plan(g, cfg, split)) {
            static const bool debug_x = getenv("NEMO_GEMM_DEBUG") != nullptr;
            if (debug_x)
                fprintf(stderr, "nemo_gemm_bf16mem M=%ld N=%ld K=%ld -> b16x %dx256 + 4 loader waves, %d K slices\n", (long)M, (long)N, (long)K,
                        b16x::tile_bm(cfg), g.split);
            const hipError_t e = b16x::launch_cfg(cfg, g, (hipStream_t)stream);
            if (e != hipSuccess) return (int32_t)e;
            NEMO_LAUNCH_CHECK();
            return NEMO_OK;
        }
    }
    return gemm_impl(2, 0, 1, M, N, K / 2, reinterpret_cast<const float*>(A), lda / 2, reinterpret_cast<const float*>(B),
                     ldb / 2, C, ldc, bias, act, m16 ? nullptr : mask, ldmask, m16 ? mask_mode - 16 : mask_mode, alpha, out_mode,
                     0, ws, ws_bytes, stream, Cb, ldcb, CbT, ldcbt,
                     m16 ? reinterpret_cast<const unsigned short*>(mask) : nullptr, ldmask, colsum, ldcs);
}

// nemo_gemm_f32 whose result is stored as bf16 (Cb) and / or as its bf16 transpose (CbT) instead of / beside C: fp32
// operands and arithmetic -- the first layer of the bf16-in-memory MLP chain (K = 105: fp32 by design), which then needs no
// cast launches behind it.  out_mode 0 only; C may be NULL.
extern "C" int32_t nemo_gemm_f32_b16out(int32_t transA, int32_t transB, int64_t M, int64_t N, int64_t K, const float* A,
                                        int64_t lda, const float* B, int64_t ldb, float* C, int64_t ldc, const float* bias,
                                        int32_t act, uint16_t* Cb, int64_t ldcb, uint16_t* CbT, int64_t ldcbt, void* ws,
                                        int64_t ws_bytes, void* stream) {
    if (!Cb && !CbT) return NEMO_EINVAL;
    return gemm_impl(0, transA, transB, M, N, K, A, lda, B, ldb, C, ldc, bias, act, nullptr, 0, 0, 1.f, 0, 0, ws, ws_bytes,
                     stream, Cb, ldcb, CbT, ldcbt);
}

// The blend-shape adjoint in fp32-EQUIVALENT split precision (round 5): C (M x N) (op)= alpha (A0 B0^T + A0 B1^T + A1 B0^T), A (M x K)
// and B (N x K) each as TWO fp16 piece planes in memory (plane p of A at A + p a_plane elements; rows k-contiguous, lda / ldb / the
// plane strides multiples of 8 elements, K even), 128 < N <= 208 -- the 64 x 208 mixed-shape tile of gemm_adj.h with its K slices dealt
// over the three plane pairs.  alpha carries the pieces' scales (1 / (s_A s_B)).
extern "C" int32_t nemo_gemm_f16x2mem_adj(int64_t M, int64_t N, int64_t K, const uint16_t* A, int64_t lda, int64_t a_plane,
                                          const uint16_t* B, int64_t ldb, int64_t b_plane, float* C, int64_t ldc, float alpha,
                                          int32_t out_mode, void* ws, int64_t ws_bytes, void* stream) {
    if (M < 0 || N <= 128 || N > 208 || K < 2 || (K & 1) || !A || !B || !C || (lda & 7) || (ldb & 7) || (a_plane & 7) || (b_plane & 7) ||
        lda < K || ldb < K || ldc < N || out_mode < 0 || out_mode > 1 || (((uintptr_t)A | (uintptr_t)B) & 15))
        return NEMO_EINVAL;
    if (M == 0) return NEMO_OK;
    const bool can_split = ws != nullptr && ws_bytes > COUNTER_BYTES && (((uintptr_t)ws) & 15) == 0;
    if (!can_split) return NEMO_EINVAL;
    const long K2 = K / 2, lda2 = lda / 2, ldb2 = ldb / 2;          // the fp32-typed view of the same bytes (pairs)
    unsigned a_bytes = 0, b_bytes = 0;
    if (!glds::extents(0, 1, M, N, K2, lda2, ldb2, &a_bytes, &b_bytes)) return NEMO_EINVAL;
    // From 1024 rows on: the 128 x 208 tile on eight waves (one workgroup per CU) -- the blend-shape tiles are fetched once per 128
    // rows; NEMO_ADJ128=0: the 64-row tile throughout (A/B aid)
    static const bool no128 = getenv("NEMO_ADJ128") != nullptr && atoi(getenv("NEMO_ADJ128")) == 0;
    const bool t128 = M >= 1024 && !no128;
    const int bm = t128 ? 128 : 64;
    const long tiles_m = (M + bm - 1) / bm;
    // slices per plane pair: ~one workgroup per CU (64-row tile: two from 9 row tiles on); every slice keeps >= 8 K tiles
    // (2400 rows, us per launch against slices per pair -- 64-row tile: 2: 227, 3: 191, 4 = 456 workgroups, this rule: 167, 5: 217,
    //  6: 219; 128-row tile: 2: 225, 3: 173, 4 = 228 workgroups, this rule: 153, 5: 205, 6: 182, 8: 164)
    int spp = (int)(((t128 || tiles_m <= 8) ? 256 : 512) / (3 * tiles_m));
    if (spp < 1) spp = 1;
    auto fits = [&](int sp) {
        return glds::adj_counter_ints(tiles_m, 3 * sp) <= COUNTER_BYTES / 4 &&
               COUNTER_BYTES + glds::adj_slab_floats(tiles_m, 3 * sp, bm) * 4 <= ws_bytes;
    };
    while (spp > 1 && ((K2 + 31) / 32 / spp < 8 || !fits(spp))) --spp;
    if (!fits(spp)) return NEMO_EINVAL;
    GemmArgs g;
    g.A = reinterpret_cast<const float*>(A); g.B = reinterpret_cast<const float*>(B); g.C = C; g.bias = nullptr; g.mask = nullptr;
    g.M = M; g.N = N; g.K = K2; g.lda = lda2; g.ldb = ldb2; g.ldc = ldc; g.ldmask = 0;
    g.act = 0; g.mask_mode = 0; g.out_mode = out_mode; g.alpha = alpha;
    g.counters = reinterpret_cast<int*>(ws);
    g.slabs = reinterpret_cast<float*>(reinterpret_cast<char*>(ws) + COUNTER_BYTES);
    long kc = (K2 + spp - 1) / spp;
    kc = ((kc + 31) / 32) * 32;
    g.k_chunk = kc;
    spp = (int)((K2 + kc - 1) / kc);
    g.split = 3 * spp;
    g.nseg = 3;
    g.seg_a[0] = 0; g.seg_a[1] = 0; g.seg_a[2] = a_plane / 2;      // (A0, B0), (A0, B1), (A1, B0)
    g.seg_b[0] = 0; g.seg_b[1] = b_plane / 2; g.seg_b[2] = 0;
    g.tiles_m = (int)tiles_m; g.tiles_n = 1; g.n_tiles = (int)tiles_m; g.t0 = 0;
    g.a_bytes = a_bytes; g.b_bytes = b_bytes; g.xcd_order = 0;
    static const bool debug_h = getenv("NEMO_GEMM_DEBUG") != nullptr;
    if (debug_h)
        fprintf(stderr, "nemo_gemm_f16x2mem_adj M=%ld N=%ld K=%ld -> 64x208 mixed-shape tile, 3 plane pairs x %d K slices\n", (long)M, (long)N, (long)K, spp);
    const hipError_t e = t128 ? glds::launch_adj128_f16x2(g, (hipStream_t)stream) : glds::launch_adj(g, (hipStream_t)stream, 2);
    if (e != hipSuccess) return (int32_t)e;
    NEMO_LAUNCH_CHECK();
    return NEMO_OK;
}

// ---- fp32-EQUIVALENT split precision for the MotionNet chain (round 6; gemm_xp.h) -------------------------------------------
// fmt 3: three bf16 pieces per operand, six piece products (no scale, no range condition); fmt 2: two fp16 pieces, three
// products, power-of-two scales carried by the caller (alpha = 1 / (s_A s_B), out_scale = the scale of the result's copies).
// Operands / copies are "xp matrices" (include/nemo_hip.h).
extern "C" int64_t nemo_xp_ld(int32_t fmt, int64_t k) { return (fmt != 2 && fmt != 3) || k < 0 ? -1 : xp::ld_for(fmt, k); }
static_assert(NEMO_XP_META_FLOATS == xp::META_FLOATS, "scale record size");

extern "C" int32_t nemo_gemm_xp(int32_t fmt, int64_t M, int64_t N, int64_t K, const uint16_t* A, int64_t lda, const uint16_t* B,
                                int64_t ldb, float* C, int64_t ldc, const float* bias, int32_t act, const uint16_t* maskx,
                                int64_t ldmask, int32_t mask_mode, float alpha, int32_t out_mode, uint16_t* Cx, int64_t ldcx,
                                uint16_t* CxT, int64_t ldcxt, float out_scale, float* colsum, int64_t ldcs, const float* metaA,
                                const float* metaB, const float* metaBias, float* metaOut, float* metaZero, void* ws, int64_t ws_bytes,
                                void* stream) {
    if ((fmt != 2 && fmt != 3) || M < 0 || N < 0 || K < 0 || !A || !B) return NEMO_EINVAL;
    if (!C && !Cx && !CxT) return NEMO_EINVAL;
    if ((((uintptr_t)A) | ((uintptr_t)B) | ((uintptr_t)Cx) | ((uintptr_t)CxT) | ((uintptr_t)maskx)) & 15) return NEMO_EINVAL;
    if (out_mode < 0 || out_mode > 1 || mask_mode < 0 || mask_mode > 2 || (mask_mode && !maskx)) return NEMO_EINVAL;
    if ((Cx && (ldcx & 7)) || (CxT && (ldcxt & 7)) || (mask_mode && (ldmask & 7)) || (C && ldc < N) || (colsum && ldcs < N)) return NEMO_EINVAL;
    if (M == 0 || N == 0) return NEMO_OK;
    if (K == 0) return NEMO_EINVAL;
    xp::Args g{};
    g.A = A; g.B = B; g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldb = ldb; g.C = C; g.ldc = ldc; g.out_mode = out_mode;
    g.bias = bias; g.act = act; g.alpha = alpha; g.maskx = mask_mode ? maskx : nullptr; g.ldmask = ldmask; g.mask_mode = mask_mode;
    g.Cx = Cx; g.ldcx = ldcx; g.CxT = CxT; g.ldcxt = ldcxt; g.out_scale = out_scale; g.colsum = colsum; g.ldcs = ldcs;
    g.metaA = fmt == 2 ? metaA : nullptr; g.metaB = fmt == 2 ? metaB : nullptr; g.metaBias = fmt == 2 ? metaBias : nullptr;
    g.metaOut = (fmt == 2 && (Cx || CxT)) ? metaOut : nullptr;
    g.metaZero = fmt == 2 ? metaZero : nullptr;
    g.counters = reinterpret_cast<int*>(ws);
    g.slabs = reinterpret_cast<float*>(reinterpret_cast<char*>(ws) + COUNTER_BYTES);
    // tile: 128 x 128 while its grid is at most ~2 rounds of the chip, 128 x 256 beyond (fewer bytes through L2 -> LDS per product)
    static const int force_bn = getenv("NEMO_XP_BN") ? atoi(getenv("NEMO_XP_BN")) : 0;
    const long t128 = ((M + 127) / 128) * ((N + 127) / 128);
    int bn = (t128 > 512 && N > 128) ? 256 : 128;
    if (force_bn == 128 || force_bn == 256) bn = force_bn;
    const long tiles = ((M + 127) / 128) * ((N + bn - 1) / bn);
    // K slices: about one workgroup per CU, every slice >= 8 K tiles of 32, at most 8 slices -- the last arriver sums a tile's slabs
    // eight at a time, one batch of round trips (2401-row chain, us per launch at 1 / 3 / 4 / 8 / 16 slices: head dW 37 / 24 / 23 /
    // 22 / 27, first-layer dX 18.6 / 16.6 / 16.8 / 19.3 / -, hidden dW 40 / 30 / 31 / 55 / -)
    static const int force_split = getenv("NEMO_XP_SPLIT") ? atoi(getenv("NEMO_XP_SPLIT")) : 0;
    int split = (int)(256 / tiles);
    if (split > 8) split = 8;
    if (split < 1) split = 1;
    while (split > 1 && (K / split < 256 || COUNTER_BYTES + tiles * split * 128L * bn * 4 > ws_bytes)) --split;
    if (force_split > 0) split = force_split;
    if (!ws || (((uintptr_t)ws) & 15) || tiles > COUNTER_BYTES / 4 || COUNTER_BYTES + tiles * split * 128L * bn * 4 > ws_bytes) split = 1;
    if (!xp::plan(g, fmt, bn, split)) return NEMO_EINVAL;
    static const bool debug_x = getenv("NEMO_GEMM_DEBUG") != nullptr;
    if (debug_x)
        fprintf(stderr, "nemo_gemm_xp fmt=%d M=%ld N=%ld K=%ld -> 128x%d, %d K slices\n", fmt, (long)M, (long)N, (long)K, bn, g.split);
    hipError_t e;
    if (fmt == 3) e = bn == 128 ? xp::launch<3, 128>(g, (hipStream_t)stream) : xp::launch<3, 256>(g, (hipStream_t)stream);
    else e = bn == 128 ? xp::launch<2, 128>(g, (hipStream_t)stream) : xp::launch<2, 256>(g, (hipStream_t)stream);
    if (e != hipSuccess) return (int32_t)e;
    NEMO_LAUNCH_CHECK();
    return NEMO_OK;
}

// Up to NEMO_GEMM_GROUP_MAX products C_i (op)= alpha_i A_i B_i^T over xp matrices in ONE launch of the 128 x 128 kernel (fp32 results,
// out_mode 0 / 1; no bias / activation / mask / copies): the four parameter gradients of the MotionNet backward.
extern "C" int32_t nemo_gemm_xp_grouped(int32_t fmt, int32_t n, const nemo_gemm_xp_problem* pr, void* ws, int64_t ws_bytes, void* stream) {
    if ((fmt != 2 && fmt != 3) || n < 0 || n > xp::MAX_GROUP || (n && !pr) || !ws || (((uintptr_t)ws) & 15)) return NEMO_EINVAL;
    xp::GroupArgs ga{};
    long tiles_tot = 0, slab_off = 0;
    int blocks = 0, m = 0;
    for (int i = 0; i < n; ++i) {
        const nemo_gemm_xp_problem& q = pr[i];
        if (q.M < 0 || q.N < 0 || q.K < 0 || !q.C || q.ldc < q.N || q.out_mode < 0 || q.out_mode > 1) return NEMO_EINVAL;
        if (q.M == 0 || q.N == 0) continue;
        if (q.K == 0 || !q.A || !q.B || ((((uintptr_t)q.A) | ((uintptr_t)q.B)) & 15)) return NEMO_EINVAL;
        xp::Args& g = ga.p[m];
        g = xp::Args{};
        g.A = q.A; g.B = q.B; g.M = q.M; g.N = q.N; g.K = q.K; g.lda = q.lda; g.ldb = q.ldb; g.C = q.C; g.ldc = q.ldc; g.out_mode = q.out_mode;
        g.alpha = q.alpha; g.out_scale = 1.f;
        g.metaA = fmt == 2 ? q.metaA : nullptr; g.metaB = fmt == 2 ? q.metaB : nullptr;
        const long tiles = ((q.M + 127) / 128) * ((q.N + 127) / 128);
        // K slices: the launch as a whole should cover the chip about twice (704 workgroups for the chain's four gradients at 2401 rows)
        int split = (int)(192 / tiles);
        if (split > 8) split = 8;
        if (split < 1) split = 1;
        while (split > 1 && q.K / split < 256) --split;
        if (!xp::plan(g, fmt, 128, split)) return NEMO_EINVAL;
        if (tiles_tot + tiles > COUNTER_BYTES / 4 || COUNTER_BYTES + (slab_off + tiles * g.split * 128L * 128) * 4 > ws_bytes) return NEMO_EINVAL;
        g.counters = reinterpret_cast<int*>(ws) + tiles_tot;
        g.slabs = reinterpret_cast<float*>(reinterpret_cast<char*>(ws) + COUNTER_BYTES) + slab_off;
        tiles_tot += tiles;
        slab_off += tiles * g.split * 128L * 128;
        ga.first[m] = blocks;
        ga.nwg[m] = (int)(tiles * (g.split > 1 ? g.split : 1));
        blocks = (blocks + ga.nwg[m] + 7) / 8 * 8;
        ++m;
    }
    ga.n = m;
    if (m == 0) return NEMO_OK;
    const hipError_t e = fmt == 3 ? xp::launch_grouped<3, 128>(ga, blocks, (hipStream_t)stream) : xp::launch_grouped<2, 128>(ga, blocks, (hipStream_t)stream);
    if (e != hipSuccess) return (int32_t)e;
    NEMO_LAUNCH_CHECK();
    return NEMO_OK;
}

extern "C" int32_t nemo_cast_xp(int32_t fmt, int32_t n, const nemo_cast_xp_desc* descs, void* stream) {
    if ((fmt != 2 && fmt != 3) || n < 0 || n > xp::MAX_CAST || (n && !descs)) return NEMO_EINVAL;
    xp::CastArgs a{};
    int tiles = 0, m = 0;
    for (int i = 0; i < n; ++i) {
        const nemo_cast_xp_desc& q = descs[i];
        if (q.rows < 0 || q.cols < 0 || !q.src || (!q.dst && !q.dstT) || q.lds < q.cols) return NEMO_EINVAL;
        if ((q.dst && (q.ldd < xp::ld_for(fmt, q.cols) || (q.ldd & 7))) || (q.dstT && (q.lddT < xp::ld_for(fmt, q.rows) || (q.lddT & 7)))) return NEMO_EINVAL;
        if (q.rows == 0 || q.cols == 0) continue;
        xp::CastDesc& d = a.d[m++];
        d.src = q.src; d.rows = q.rows; d.cols = q.cols; d.lds = q.lds; d.dst = q.dst; d.ldd = q.ldd; d.dstT = q.dstT; d.lddT = q.lddT;
        d.scale = fmt == 2 ? q.scale : 1.f;
        d.meta = fmt == 2 ? q.meta : nullptr;
        d.tile0 = tiles; d.tiles_c = nemo_cdiv(q.cols, 32);
        tiles += d.tiles_c * nemo_cdiv(q.rows, 32);
    }
    a.n = m;
    if (m == 0) return NEMO_OK;
    if (fmt == 3) hipLaunchKernelGGL(xp::cast_xp_kernel<3>, dim3(tiles), dim3(256), 0, (hipStream_t)stream, a);
    else hipLaunchKernelGGL(xp::cast_xp_kernel<2>, dim3(tiles), dim3(256), 0, (hipStream_t)stream, a);
    NEMO_LAUNCH_CHECK();
    return NEMO_OK;
}

extern "C" int32_t nemo_absmax_multi(int32_t n, const nemo_absmax_desc* descs, void* stream) {
    xp::AbsmaxArgs a;
    int blocks = 0;
    const int rc = xp::absmax_fill(a, n, descs, blocks);
    if (rc) return rc;
    if (a.n == 0) return NEMO_OK;
    hipLaunchKernelGGL(xp::absmax_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, a);
    NEMO_LAUNCH_CHECK();
    return NEMO_OK;
}

// rows of the `colsum` scratch nemo_gemm_bf16mem fills for an M-row result: one per 32-row band of its 64 x 64 tiles
extern "C" int64_t nemo_gemm_colsum_rows(int64_t M) { return M < 0 ? -1 : 2 * ((M + 63) / 64); }

namespace {
// dst (bf16) = src (fp32), optionally transposed; 32 x 32 tiles through LDS so that both sides move whole lines.
// Columns (transpose: rows) beyond the source up to `pad_to` are zero-filled: k-pads of a GEMM operand.
__global__ __launch_bounds__(256) void cast_bf16_kernel(long rows, long cols, const float* __restrict__ src, long lds,
                                                        unsigned short* __restrict__ dst, long ldd, int transpose, long pad_to) {
    __shared__ float t[32][33];
    const long r0 = (long)blockIdx.y * 32, c0 = (long)blockIdx.x * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int i = ty; i < 32; i += 8) {
        const long r = r0 + i, c = c0 + tx;
        t[i][tx] = (r < rows && c < cols) ? src[r * lds + c] : 0.f;
    }
    __syncthreads();
    if (!transpose) {
        for (int i = ty; i < 32; i += 8) {
            const long r = r0 + i, c = c0 + tx;
            if (r < rows && c < pad_to) dst[r * ldd + c] = __builtin_bit_cast(unsigned short, (__bf16)t[i][tx]);
        }
    } else {
        for (int i = ty; i < 32; i += 8) {
            const long c = c0 + i, r = r0 + tx;                  // dst row = source column
            if (c < cols && r < pad_to) dst[c * ldd + r] = __builtin_bit_cast(unsigned short, (__bf16)t[tx][i]);
        }
    }
}
}  // namespace

extern "C" int32_t nemo_cast_bf16(int64_t rows, int64_t cols, const float* src, int64_t lds, uint16_t* dst, int64_t ldd,
                                  int32_t transpose, void* stream) {
    if (rows < 0 || cols < 0 || !src || !dst || lds < cols || ldd < (transpose ? rows : cols)) return NEMO_EINVAL;
    if (rows == 0 || cols == 0) return NEMO_OK;
    const long inner = transpose ? rows : cols;
    long pad_to = (inner + 7) / 8 * 8;
    if (pad_to > ldd) pad_to = ldd;
    dim3 grid(nemo_cdiv(transpose ? cols : pad_to, 32), nemo_cdiv(transpose ? pad_to : rows, 32));
    hipLaunchKernelGGL(cast_bf16_kernel, grid, dim3(256), 0, (hipStream_t)stream, (long)rows, (long)cols, src, (long)lds,
                       dst, (long)ldd, (int)transpose, pad_to);
    NEMO_LAUNCH_CHECK();
    return NEMO_OK;
}

namespace {
// dst row = [hi | lo | hi] (order 0) or [hi | hi | lo] (order 1), hi = bf16(x), lo = bf16(x - hi); segments of `seg` columns
__global__ __launch_bounds__(256) void cast_bf16_split3_kernel(long rows, long cols, long seg, const float* __restrict__ src, long lds,
                                                               unsigned short* __restrict__ dst, long ldd, int order) {
    const long c = (long)blockIdx.x * 64 + (threadIdx.x & 63);
    const long r = (long)blockIdx.y * 4 + (threadIdx.x >> 6);
    if (r >= rows || c >= seg) return;
    const float x = c < cols ? src[r * lds + c] : 0.f;
    const __bf16 hi = (__bf16)x;
    const __bf16 lo = (__bf16)(x - (float)hi);
    const unsigned short h = __builtin_bit_cast(unsigned short, hi), l = __builtin_bit_cast(unsigned short, lo);
    unsigned short* d = dst + r * ldd + c;
    d[0] = h;
    d[seg] = order ? h : l;
    d[2 * seg] = order ? l : h;
}
}  // namespace

extern "C" int32_t nemo_cast_bf16_split3(int64_t rows, int64_t cols, const float* src, int64_t lds, uint16_t* dst, int64_t ldd,
                                         int32_t order, void* stream) {
    const long seg = (cols + 7) / 8 * 8;
    if (rows < 0 || cols < 0 || !src || !dst || lds < cols || ldd < 3 * seg || (order != 0 && order != 1)) return NEMO_EINVAL;
    if (rows == 0 || cols == 0) return NEMO_OK;
    dim3 grid(nemo_cdiv(seg, 64), nemo_cdiv(rows, 4));
    hipLaunchKernelGGL(cast_bf16_split3_kernel, grid, dim3(256), 0, (hipStream_t)stream, (long)rows, (long)cols, seg, src, (long)lds,
                       dst, (long)ldd, (int)order);
    NEMO_LAUNCH_CHECK();
    return NEMO_OK;
}

// Several independent products of ONE layout in one launch (glds::gemm_glds_grouped_kernel): the parameter gradients
// dW_l = dY_l^T X_l of the whole MLP backward.  Anything the grouped kernel does not cover (mixed layouts, unaligned
// operands, more than MAX_GROUP problems, tuning overrides) runs as consecutive single launches: same results.
static int32_t gemm_grouped_impl(bool bf16, int32_t n, const nemo_gemm_problem* pr, void* ws, int64_t ws_bytes, void* stream) {
    if (n < 0 || (n && !pr) || ws_bytes < 0 || (ws_bytes > 0 && !ws)) return NEMO_EINVAL;
    for (int i = 0; i < n; ++i) {
        const nemo_gemm_problem& q = pr[i];
        if (q.M < 0 || q.N < 0 || q.K < 0 || !q.C || (q.out_mode != 0 && q.out_mode != 1)) return NEMO_EINVAL;
        if (q.M && q.N && q.K && (!q.A || !q.B)) return NEMO_EINVAL;
    }
    const bool can_split = ws != nullptr && ws_bytes > COUNTER_BYTES && (((uintptr_t)ws) & 15) == 0;
    bool ok = n >= 2 && n <= glds::MAX_GROUP && !getenv("NEMO_GEMM_TILE");
    glds::GroupArgs ga;
    long total_tiles = 0, small_tiles = 0;
    for (int i = 0; ok && i < n; ++i) {
        const nemo_gemm_problem& q = pr[i];
        ok = q.M > 0 && q.N > 0 && q.K > 0 && q.transA == pr[0].transA && q.transB == pr[0].transB &&
             aligned16(q.A, q.lda) && aligned16(q.B, q.ldb);
        if (!ok) break;
        GemmArgs& g = ga.p[i];
        g = GemmArgs{};
        ok = glds::extents(q.transA, q.transB, q.M, q.N, q.K, q.lda, q.ldb, &g.a_bytes, &g.b_bytes);
        g.A = q.A; g.B = q.B; g.C = q.C; g.bias = nullptr; g.mask = nullptr;
        g.M = q.M; g.N = q.N; g.K = q.K; g.lda = q.lda; g.ldb = q.ldb; g.ldc = q.ldc; g.ldmask = 0;
        g.act = 0; g.mask_mode = 0; g.out_mode = q.out_mode; g.alpha = q.alpha;
        g.tiles_m = (int)((q.M + 63) / 64); g.tiles_n = (int)((q.N + 63) / 64); g.n_tiles = g.tiles_m * g.tiles_n;
        g.xcd_order = 0;
        total_tiles += g.n_tiles;
        if (g.n_tiles <= 64) small_tiles += g.n_tiles;
    }
    if (!ok) {
        for (int i = 0; i < n; ++i) {
            const nemo_gemm_problem& q = pr[i];
            const int32_t rc = gemm_impl(bf16, q.transA, q.transB, q.M, q.N, q.K, q.A, q.lda, q.B, q.ldb, q.C, q.ldc, nullptr, 0,
                                         nullptr, 0, 0, q.alpha, q.out_mode, 0, ws, ws_bytes, stream);
            if (rc) return rc;
        }
        return NEMO_OK;
    }
    // Balance: three blocks are resident per CU (768 slots).  The large problems keep whole tiles; if slots are left over in
    // the launch's only round, the SMALL problems (<= 64 tiles: the head and first-layer gradients) are cut along K to fill
    // them -- 512 + 48 + 32 tiles of the 8 x 300 step become 512 whole tiles + 80 tiles x 3 slices = 752 blocks.
    int S = 1;
    if (can_split && small_tiles > 0 && total_tiles < 768) {
        S = 1 + (int)((768 - total_tiles) / small_tiles);
        if (S > 4) S = 4;
    }
    long ticket0 = 0, slab_bytes = COUNTER_BYTES, blk = 0;
    for (int i = 0; i < n; ++i) {
        GemmArgs& g = ga.p[i];
        int s_i = (g.n_tiles <= 64) ? S : 1;
        while (s_i > 1 && g.K / s_i < 256) --s_i;                         // >= 8 K tiles per slice
        long kc = ((g.K + s_i - 1) / s_i + 31) / 32 * 32;
        g.k_chunk = kc;
        g.split = (int)((g.K + kc - 1) / kc);
        g.t0 = g.split > 1 ? 0 : g.n_tiles;
        g.counters = reinterpret_cast<int*>(ws) + ticket0;
        g.slabs = reinterpret_cast<float*>(reinterpret_cast<char*>(ws) + slab_bytes);
        if (g.split > 1) {
            ticket0 += g.n_tiles;
            slab_bytes += (long)g.n_tiles * g.split * 64 * 64 * 4;
            if (ticket0 > COUNTER_BYTES / 4 || slab_bytes > ws_bytes) return NEMO_EINVAL;
        }
        ga.blk0[i] = (int)blk;
        blk += g.split > 1 ? (long)g.n_tiles * g.split : g.n_tiles;
    }
    for (int i = n; i <= glds::MAX_GROUP; ++i) ga.blk0[i] = (int)blk;
    for (int i = n; i < glds::MAX_GROUP; ++i) ga.p[i] = ga.p[0];
    static const bool debug = getenv("NEMO_GEMM_DEBUG") != nullptr;
    if (debug) fprintf(stderr, "nemo_gemm_grouped n=%d ta=%d tb=%d: %ld tiles, small-problem split %d -> %ld blocks\n", n,
                       pr[0].transA, pr[0].transB, total_tiles, S, blk);
    const bool akc = !pr[0].transA, bkc = pr[0].transB != 0;
    hipStream_t s = (hipStream_t)stream;
    hipError_t e;
    if (bf16) {
        if (akc && bkc) e = glds::launch_grouped<64, 64, 32, 32, 32, true, true, 3, true, true>(ga, s);
        else if (akc) e = glds::launch_grouped<64, 64, 32, 32, 32, true, false, 3, true, true>(ga, s);
        else if (bkc) e = glds::launch_grouped<64, 64, 32, 32, 32, false, true, 3, true, true>(ga, s);
        else e = glds::launch_grouped<64, 64, 32, 32, 32, false, false, 3, true, true>(ga, s);
    } else if (akc && bkc) e = glds::launch_grouped<64, 64, 32, 32, 32, true, true, 3, true>(ga, s);
    else if (akc) e = glds::launch_grouped<64, 64, 32, 32, 32, true, false, 3, true>(ga, s);
    else if (bkc) e = glds::launch_grouped<64, 64, 32, 32, 32, false, true, 3, true>(ga, s);
    else e = glds::launch_grouped<64, 64, 32, 32, 32, false, false, 3, true>(ga, s);
    if (e != hipSuccess) return (int32_t)e;
    NEMO_LAUNCH_CHECK();
    return NEMO_OK;
}

extern "C" int32_t nemo_gemm_grouped_f32(int32_t n, const nemo_gemm_problem* problems, void* ws, int64_t ws_bytes, void* stream) {
    return gemm_grouped_impl(false, n, problems, ws, ws_bytes, stream);
}
extern "C" int32_t nemo_gemm_grouped_bf16(int32_t n, const nemo_gemm_problem* problems, void* ws, int64_t ws_bytes, void* stream) {
    return gemm_grouped_impl(true, n, problems, ws, ws_bytes, stream);
}

// Column sums of a row-major (M x N) matrix: out[n] (+)= sum_m X[m][n].  Bias gradients.
namespace {
__global__ __launch_bounds__(256) void colsum_kernel(const float* __restrict__ X, long M, long N, long ldx,
                                                     float* __restrict__ out, long rows_per_block, NemoRed rr) {
    const long n = (long)blockIdx.x * 64 + (threadIdx.x & 63);
    const long mbeg = (long)blockIdx.y * rows_per_block;
    const long mend = min(M, mbeg + rows_per_block);
    float s = 0.f;
    if (n < N)
        for (long m = mbeg + (threadIdx.x >> 6); m < mend; m += 4) s += X[m * ldx + n];
    __shared__ float red[4][64];
    __shared__ int rflag;
    red[threadIdx.x >> 6][threadIdx.x & 63] = s;
    __syncthreads();
    const float t = threadIdx.x < 64 ? red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x] : 0.f;
    nemo_colsum_finish(t, n, N, out, rr, (int)blockIdx.x, (int)gridDim.x, (int)blockIdx.y, (int)gridDim.y, (int)blockIdx.x, &rflag);
}
}  // namespace

namespace {
struct ColsumBatch { nemo_colsum_desc d[NEMO_COLSUM_MAX]; };
__global__ __launch_bounds__(256) void colsum_multi_kernel(ColsumBatch b, long rows_per_block, NemoRed rr) {
    const nemo_colsum_desc d = b.d[blockIdx.z];
    const long n = (long)blockIdx.x * 64 + (threadIdx.x & 63);
    const long mbeg = (long)blockIdx.y * rows_per_block;
    if ((long)blockIdx.x * 64 >= d.N || d.M <= 0) return;           // block-uniform
    // (a matrix shorter than the launch's tallest: its row chunks beyond M deposit zeros -- every strip sees gridDim.y arrivals)
    const long mend = min((long)d.M, mbeg + rows_per_block);
    float s = 0.f;
    if (n < d.N)
        for (long m = mbeg + (threadIdx.x >> 6); m < mend; m += 4) s += d.X[m * d.ldx + n];
    __shared__ float red[4][64];
    __shared__ int rflag;
    red[threadIdx.x >> 6][threadIdx.x & 63] = s;
    __syncthreads();
    const float t = threadIdx.x < 64 ? red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x] : 0.f;
    const int strip = (int)(blockIdx.z * gridDim.x + blockIdx.x);
    nemo_colsum_finish(t, n, d.N, d.out, rr, strip, (int)(gridDim.x * gridDim.z), (int)blockIdx.y, (int)gridDim.y, strip, &rflag);
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
    hipLaunchKernelGGL(colsum_multi_kernel, grid, dim3(256), 0, (hipStream_t)stream, b, rows_per_block,
                       nemo_red_take((size_t)grid.x * grid.y * grid.z * 64, (int)(grid.x * grid.z)));
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
                       (long)ldx, out, rows_per_block, nemo_red_take((size_t)grid.x * grid.y * 64, (int)grid.x));
    NEMO_LAUNCH_CHECK();
    return NEMO_OK;
}
