// Split-precision large-tile GEMM for the fp32 MotionNet chain (round 6):
//   C (M x N) (op)= epilogue(alpha * A B^T)  in fp32-EQUIVALENT arithmetic on the 16-bit matrix cores.
// Replaces, for fp32 builds (args.mlp_gemm = 'f32_split'), the nn.Linear products of MotionNet and their autograd
// (nemo/neural_motion_model.py:58-71, :130-148): forward Y = X W^T, activation gradient dX = dY W, parameter gradient
// dW = dY^T X -- the same three roles gemm_b16x.h plays for args.gemm_dtype = 'bf16', same operand convention (both operands
// k-contiguous "copies", the producing launch's epilogue writes the copies the next launches read).
//
// Arithmetic.  Every fp32 operand x is held as NP 16-bit pieces whose sum is x:
//   NP = 3, bf16:  x0 = bf16(x), x1 = bf16(x - x0), x2 = bf16(x - x0 - x1)   -- 8 + 8 + 8 significant bits = x EXACTLY, bf16 has
//                  fp32's exponent range: no scale, no range condition;
//   NP = 2, fp16:  x0 = fp16(s x), x1 = fp16(s x - x0) with a power-of-two s  -- 11 + 11 bits + the remainder's sign; needs
//                  |s x| < 65504 (the caller's range guard) and loses relative precision below |s x| ~ 2^-3.
// A product keeps the piece products of weight >= 2^-24:  NP = 3: a0 b0 + (a0 b1 + a1 b0 + a1 b1 + a0 b2 + a2 b0), the dropped
// terms are <= 2^-25 |a b|;  NP = 2: a0 b0 + (a0 b1 + a1 b0).  Each piece product is exact in fp32; accumulation in fp32 in TWO
// accumulators per output (the leading product / the minor ones: the minor sum is ~2^-8 of the result, so its roundings do not
// count, and the leading accumulator takes K / 16 roundings where the fp32 MFMA kernel's takes K / 2).
//
// Memory format of an operand ("xp matrix", rows x K): row r, k-block kb = k / 32, piece p, element k % 32 at 16-bit index
//   r * ld + kb * 32 NP + p * 32 + (k & 31),          ld = 32 NP ceil(K / 32) (or more),
// i.e. the NP pieces of 32 consecutive k of a row are 64 NP consecutive bytes: ONE ring stage of the kernel (BK = 32) moves whole
// 128- / 192-byte row segments, the granularity at which a CU's L2 -> LDS stream reaches its full rate (64-byte segments: half,
// profiles/r05_dma_rate.txt).  Elements k in [K, 32 ceil(K / 32)) of every row are ZERO in both operands (cast kernel and
// epilogues write them so): the kernel never masks inside a k-block.
//
// Kernel: 128 x BN tile (BN = 128 / 256), 8 MFMA waves (2 x 4; wave tile 64 x BN / 4 as 32 x 32 accumulators of
// v_mfma_f32_32x32x16_{bf16,f16}) + 4 loader waves that issue every LDS-DMA piece (gemm_b16x.h's scheme).  LDS image of a
// stage: [row][4 NP + 1 chunks of 16 B] -- the odd row stride (13 / 9 chunks) makes every ds_read_b128 lane group (16 distinct
// rows mod 16, one chunk index) conflict-free without a swizzle; the pad chunk is "fetched" out of bounds (nothing moves).
// Rows beyond M / N read as zeros through the descriptors' extents.  Split-K: write-through slabs + ticket + ordered
// last-arriver sum (deterministic), as gemm_b16x.h.
//
// Epilogue: v = maskfn(act(alpha * acc + bias)); outputs (any subset): C fp32 (store / +=), Cx = v as an xp matrix [m][n],
// CxT = v^T as an xp matrix [n][m] (both: pieces of v * out_scale), colsum = per-32-row-band column sums of v (the layer's bias
// gradient, one writer per element).  The ReLU' mask is read from piece 0 of an xp copy of the activation (sign and zero of
// piece 0 are those of the value).
#pragma once
#include <type_traits>
#include "common.h"
#include "gemm_glds.h"
#include "absmax.h"

namespace xp {

using nemo_meta::META_FLOATS;
using nemo_meta::META_SLOTS;
using nemo_meta::meta_absmax;
using nemo_meta::meta_absmax_put;

using glds::f32x16;
using glds::i32x4;
typedef __bf16 xbf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 xf16x8 __attribute__((ext_vector_type(8)));

struct Args {
    const unsigned short* A; const unsigned short* B;       // xp matrices (16-bit patterns)
    long M, N, K;
    long lda, ldb;                                          // 16-bit elements; multiples of 8, >= 32 NP ceil(K / 32)
    float* C; long ldc; int out_mode;                       // C may be NULL; 0 store, 1 +=
    const float* bias; int act; float alpha;                // act 0 / 1 ReLU / 2 LeakyReLU(0.01)
    const unsigned short* maskx; long ldmask; int mask_mode;      // 0 none, 1: v = piece0(mask[m][n]) > 0 ? v : 0, 2: ... : 0.01 v
    unsigned short* Cx; long ldcx;                          // >= 32 NP ceil(N / 32)
    unsigned short* CxT; long ldcxt;                        // >= 32 NP ceil(M / 32)
    float out_scale;                                        // pieces of v * out_scale (1 for bf16 pieces; fmt 2 without metaOut)
    // fmt 2 (fp16 pieces): device-resident scale records, float[META_FLOATS] = {power-of-two scale s of the pieces, -, 32 absmax slots of
    // the TRUE values (the absmax is their maximum: same-address atomics of ~1200 waves serialise, 15 us per launch)}.
    // metaA / metaB (may be NULL: scale 1): the operands' records; alpha is divided by s_A s_B in the kernel.  metaOut (may be NULL):
    // the record of the result's copies -- the kernel derives s_out = 2^floor(log2(2^15 / bound)) from the BOUND
    // |alpha| K absmax_A absmax_B + absmax_bias >= |v| (no overflow by construction), writes it to metaOut[0] and accumulates the
    // result's true absmax into metaOut's slots (atomic max over non-negative floats; zero before the launch).
    const float* metaA; const float* metaB; const float* metaBias; float* metaOut;
    float* metaZero;                                        // may be NULL: a record whose absmax slots this launch returns to zero (it is
                                                            // stream-ordered behind their last reader; its scale [0] stays)
    float* colsum; long ldcs;                               // rows: one per 32-row band, 2 ceil(M / 64) of them
    float* slabs; int* counters;
    long k_chunk; int split;                                // K range per slice (multiple of 32)
    int tiles_m, tiles_n;
    unsigned a_bytes, b_bytes, mask_bytes;                  // buffer extents
};

template <int NP, int BN>
struct Geo {
    static_assert(NP == 2 || NP == 3, "two fp16 or three bf16 pieces");
    static_assert(BN == 128 || BN == 256, "128 x 128 or 128 x 256");
    static constexpr int BM = 128, WGM = 2, WGN = 4, LW = 4;
    static constexpr int WM = BM / WGM, WN = BN / WGN, TM = WM / 32, TN = WN / 32;
    static constexpr int CH = 4 * NP;                       // 16-byte chunks of one k-block of one row (all pieces)
    static constexpr int CPR = CH + 1;                      // ... + the pad chunk: LDS row stride in chunks (odd)
    static constexpr int RS = CPR * 16;                     // LDS row stride, bytes
    static constexpr int KBB = CH * 16;                     // bytes of a row's k-block in memory
    static constexpr int ROWS = BM + BN;
    static constexpr int STAGE = ROWS * RS;
    static constexpr int NPC = ROWS * CPR / 64;             // 1-KiB DMA pieces per stage
    static_assert(ROWS * CPR % 64 == 0 && BM * CPR % 64 == 0, "pieces do not straddle the operands");
    static constexpr int NPA = BM * CPR / 64;
    static constexpr int GW = (NPC + LW - 1) / LW;          // pieces per loader wave and stage
    static constexpr bool DUMMY = GW * LW != NPC;
    static constexpr int NST = (163840 - (DUMMY ? 1024 : 0)) / STAGE >= 4 ? 4 : (163840 - (DUMMY ? 1024 : 0)) / STAGE;
    static_assert(NST >= 2, "at least a double buffer");
    static_assert((NST - 1) * GW < 64, "vmcnt is a 6-bit counter");
    static constexpr int RING = NST * STAGE;
    static constexpr int THREADS = 64 * (8 + LW);
    // epilogue scratch per wave, one 32-column block j at a time: [m 64][k-block + 16 B] or [n 32][TM k-blocks + 16 B]
    static constexpr int SROW = KBB + 16, TROW = TM * KBB + 16;
    static constexpr int EPI = (WM * SROW > 32 * TROW ? WM * SROW : 32 * TROW);
    static constexpr int LDS = (RING + (DUMMY ? 1024 : 0) > 8 * EPI ? RING + (DUMMY ? 1024 : 0) : 8 * EPI);
    static_assert(LDS <= 163840, "160 KiB of LDS per workgroup");
};

// the power of two s with s * bound in [2^(top - 1), 2^top), exponent clamped to [-120, 120] (only ever towards SMALLER |s x|: a
// clamped scale costs precision, never range); bound <= 0 or not finite: 1
__device__ __forceinline__ float pow2_scale(float bound, int top) {
    if (!(bound > 0.f) || !(bound < 3.0e38f)) return 1.f;
    int e;
    (void)frexpf(bound, &e);                    // bound = m 2^e, m in [0.5, 1)
    int k = top - e;
    k = k > 120 ? 120 : (k < -120 ? -120 : k);
    return ldexpf(1.f, k);
}

constexpr int OOB = (int)0x80000000u;      // a buffer offset beyond every descriptor of this kernel (extents < 2^31)

// the NP pieces of v (already multiplied by the output scale) as 16-bit patterns
template <int NP>
__device__ __forceinline__ void split_pieces(float v, unsigned short (&out)[NP]) {
    if constexpr (NP == 3) {
        const __bf16 x0 = (__bf16)v;
        const float r1 = v - (float)x0;
        const __bf16 x1 = (__bf16)r1;
        const __bf16 x2 = (__bf16)(r1 - (float)x1);
        out[0] = __builtin_bit_cast(unsigned short, x0);
        out[1] = __builtin_bit_cast(unsigned short, x1);
        out[2] = __builtin_bit_cast(unsigned short, x2);
    } else {
        const _Float16 h0 = (_Float16)v;
        const _Float16 h1 = (_Float16)(v - (float)h0);
        out[0] = __builtin_bit_cast(unsigned short, h0);
        out[1] = __builtin_bit_cast(unsigned short, h1);
    }
}

template <int NP>
__device__ __forceinline__ f32x16 mfma16(i32x4 a, i32x4 b, f32x16 c) {
    if constexpr (NP == 3) return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(xbf16x8, a), __builtin_bit_cast(xbf16x8, b), c, 0, 0, 0);
    else return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(xf16x8, a), __builtin_bit_cast(xf16x8, b), c, 0, 0, 0);
}

// The kernel body: workgroup `bid` of the `nwg` of problem g (a launch of its own, or one of a grouped launch's problems)
template <int NP, int BN>
__device__ __forceinline__ void gemm_xp_body(const Args& g, const int bid, const int nwg) {
    using Q = Geo<NP, BN>;
    constexpr int BM = Q::BM, WM = Q::WM, WN = Q::WN, TM = Q::TM, TN = Q::TN, GW = Q::GW, NPA = Q::NPA, NPC = Q::NPC, NST = Q::NST;
    constexpr int CH = Q::CH, CPR = Q::CPR, RS = Q::RS, KBB = Q::KBB, LW = Q::LW, WGN = Q::WGN;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];          // the ONLY LDS object of the kernel

    // ---- block -> (K slice, row tile, column tile); every XCD (block b runs on XCD b % 8) gets a contiguous run of the
    // (slice, tm, tn) order with tn fastest: its blocks share A row panels in its L2 (gemm_b16x.h)
    const int xq = nwg >> 3, xr = nwg & 7, xcd = bid & 7;
    const int lin = (xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq) + (bid >> 3);
    const int tn = lin % g.tiles_n, tm = (lin / g.tiles_n) % g.tiles_m, slice = lin / (g.tiles_n * g.tiles_m);
    const int tile = tn * g.tiles_m + tm;
    const long m0 = (long)tm * BM, n0 = (long)tn * BN;
    const long kbeg = (long)slice * g.k_chunk;
    const long kend = g.split > 1 ? min(g.K, kbeg + g.k_chunk) : g.K;
    const int klen = (int)(kend > kbeg ? kend - kbeg : 0);
    const int nt = (klen + 31) / 32;                        // >= 1: every slice is non-empty by construction

    const int lane = threadIdx.x & 63;
    const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    __builtin_assume(wid >= 0 && wid < 8 + LW);
    const bool loader = wid >= 8;                           // (wave-uniform)
    const int wm = wid / WGN, wn = wid % WGN;
    const int l31 = lane & 31, lh = lane >> 5;

    // acc[0]: the leading piece product a0 b0;  acc[1]: the minor ones (128 x 256: one set -- two do not fit 168 registers)
    constexpr int NACC = TN == 1 ? 2 : 1;
    f32x16 acc[NACC][TM][TN];
#pragma unroll
    for (int h = 0; h < NACC; ++h)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[h][i][j][r] = 0.f;

    // fmt 2: the operands' scales leave through alpha; the scale of the result's copies from the bound on |v| (read here, at the top:
    // five dependent global loads in front of the epilogue cost ~5 us per launch)
    float alpha = g.alpha, out_scale = g.out_scale;
    if constexpr (NP == 2) {
        if (!loader) {
            const float sA = g.metaA ? g.metaA[0] : 1.f, sB = g.metaB ? g.metaB[0] : 1.f;
            alpha = (g.alpha / sA) / sB;
            if (g.metaOut) {
                const float mA = g.metaA ? meta_absmax(g.metaA) : 1.f, mB = g.metaB ? meta_absmax(g.metaB) : 1.f;
                const float bound = fabsf(g.alpha) * (float)g.K * mA * mB + (g.metaBias ? meta_absmax(g.metaBias) : 0.f);
                out_scale = pow2_scale(bound, 15);
            }
        }
    }

    const unsigned smem_byte = (unsigned)reinterpret_cast<unsigned long long>(smem);

    if (loader) {
        // ---- LDS-DMA.  Lane l of piece q owns chunk index gq = 64 q + l of the stage's [row][CPR] grid: row gq / CPR, chunk
        // gq % CPR (the pad chunk: out of bounds, nothing is fetched); it lands at byte 16 gq of the stage = row * RS + 16 chunk.
        // Loader w issues pieces w, w + 4, ...; pieces beyond the stage (q >= NPC) are dummies into the 1 KiB behind the ring:
        // every loader has the same number of DMAs in flight per tile, which the counted vmcnt relies on.
        const int iw = wid - 8;
        auto rsrc = [](const void* p, unsigned bytes) {
            const unsigned long long b = reinterpret_cast<unsigned long long>(p);
            return i32x4{(int)(unsigned)b, (int)(unsigned)((b >> 32) & 0xffffu), (int)bytes, 0x00020000};
        };
        const i32x4 rsA = rsrc(g.A, g.a_bytes), rsB = rsrc(g.B, g.b_bytes);
        const unsigned baseA = (unsigned)((m0 * g.lda) * 2 + (kbeg / 32) * KBB), baseB = (unsigned)((n0 * g.ldb) * 2 + (kbeg / 32) * KBB);
        i32x4 prs[GW];
        unsigned plds[GW], psrc[GW];
        int pvoff[GW];
#pragma unroll
        for (int p = 0; p < GW; ++p) {
            const int q = iw + LW * p;
            const bool isA = q < NPA, dummy = q >= NPC;
            const int gq = q * 64 + lane, row = gq / CPR, c = gq - row * CPR;
            prs[p] = isA ? rsA : rsB;
            plds[p] = dummy ? (unsigned)Q::RING : (unsigned)(q * 1024);
            psrc[p] = isA ? baseA : baseB;
            const long rl = isA ? (long)row * g.lda : (long)(row - BM) * g.ldb;
            pvoff[p] = (dummy || c == CH) ? OOB : (int)(rl * 2 + c * 16);
        }
        auto dma = [&](int p, int t, int stage) {
            const unsigned st = plds[p] == (unsigned)Q::RING ? 0u : (unsigned)(stage * Q::STAGE);
            glds::dma_piece(prs[p], smem_byte + st + plds[p], t < nt ? pvoff[p] : OOB, psrc[p] + (unsigned)t * KBB);
        };
#pragma unroll
        for (int t = 0; t < NST; ++t)
#pragma unroll
            for (int p = 0; p < GW; ++p) dma(p, t, t);
        glds::wait_vmcnt<(NST - 1) * GW>();
        asm volatile("" ::: "memory");
        __builtin_amdgcn_s_barrier();
        int cur = 0;
        for (int t = 0; t < nt - 1; ++t) {
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NST - 2) * GW) : "memory");
            __builtin_amdgcn_s_barrier();
#pragma unroll
            for (int p = 0; p < GW; ++p) dma(p, t + NST, cur);
            cur = cur + 1 == NST ? 0 : cur + 1;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // (pieces of the tiles behind the slice: zeros into dead stages)
    } else {
        // ---- operand fetch: k-step s (0 / 1) of a stage, lane half lh, piece p: chunk 4 p + 2 s + lh of row l31 of a 32-row block
        struct Frag { i32x4 a[TM][NP], b[TN][NP]; };
        const unsigned fa0 = (unsigned)((wm * WM + l31) * RS + lh * 16), fb0 = (unsigned)((BM + wn * WN + l31) * RS + lh * 16);
        auto fetch = [&](Frag& f, int stage, int s) {
            const unsigned char* st = smem + stage * Q::STAGE + s * 32;
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int p = 0; p < NP; ++p) f.a[i][p] = *reinterpret_cast<const i32x4*>(st + fa0 + i * 32 * RS + p * 64);
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int p = 0; p < NP; ++p) f.b[j][p] = *reinterpret_cast<const i32x4*>(st + fb0 + j * 32 * RS + p * 64);
        };
        // the MFMAs of one k-step: minor products first (smallest weights first), the leading one last; `after_first()` runs
        // behind the first MFMA (the next fragment reads are issued there, gemm_b16x.h)
        auto mma = [&](const Frag& f, auto&& after_first) {
            constexpr int NPR = NP == 3 ? 6 : 3;
            constexpr int pa[6] = {1, 0, 2, 0, 1, 0}, pb[6] = {1, 2, 0, 1, 0, 0};         // NP = 3
            constexpr int qa[3] = {0, 1, 0}, qb[3] = {1, 0, 0};                            // NP = 2
            bool first = true;
#pragma unroll
            for (int u = 0; u < NPR; ++u) {
                const int ia = NP == 3 ? pa[u] : qa[u], ib = NP == 3 ? pb[u] : qb[u];
                const int h = (u == NPR - 1 || NACC == 1) ? 0 : 1;
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j) {
                        acc[h][i][j] = mfma16<NP>(f.a[i][ia], f.b[j][ib], acc[h][i][j]);
                        if (first) {
                            first = false;
                            __builtin_amdgcn_sched_barrier(0);
                            after_first();
                            __builtin_amdgcn_sched_barrier(0);
                        }
                    }
            }
        };
        Frag F, G;
        asm volatile("" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        fetch(F, 0, 0);
        int cur = 0;                                        // stage of tile t
        for (int t = 0; t < nt - 1; ++t) {
            mma(F, [&] { fetch(G, cur, 1); });
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            const int nxt = cur + 1 == NST ? 0 : cur + 1;
            mma(G, [&] { fetch(F, nxt, 0); });
            __builtin_amdgcn_sched_barrier(0);
            cur = nxt;
        }
        mma(F, [&] { fetch(G, cur, 1); });
        __builtin_amdgcn_sched_barrier(0);
        mma(G, [] {});
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if constexpr (NACC == 2)
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[0][i][j][r] += acc[NACC - 1][i][j][r];
    }

    // ---- split-K: publish the partial tile write-through, take a ticket; the last arriver sums all slices in slice order
    // (deterministic) and runs the epilogue.  Slab layout = the register image: float4 #(i, j, r4) of thread t.
    constexpr int NV4 = TM * TN * 4;
    if (g.split > 1) {
        float4* slab = reinterpret_cast<float4*>(g.slabs) + ((size_t)tile * g.split + slice) * (size_t)(BM * BN / 4);
        if (!loader)
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int r4 = 0; r4 < 4; ++r4) {
                        const glds::f32x4 vv = {acc[0][i][j][4 * r4], acc[0][i][j][4 * r4 + 1], acc[0][i][j][4 * r4 + 2], acc[0][i][j][4 * r4 + 3]};
                        float4* dst = slab + ((i * TN + j) * 4 + r4) * 512 + threadIdx.x;
                        asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(dst), "v"(vv) : "memory");
                    }
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __syncthreads();                                    // (also: every wave is done reading the ring)
        int* flag = reinterpret_cast<int*>(smem);
        if (threadIdx.x == 0)
            *flag = __hip_atomic_fetch_add(g.counters + tile, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
        const int ticket = *flag;
        if (ticket != g.split - 1) return;
        if (threadIdx.x == 0) {
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            __hip_atomic_store(g.counters + tile, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ready for the next launch
        }
        __syncthreads();
        const float4* base = reinterpret_cast<const float4*>(g.slabs) + (size_t)tile * g.split * (size_t)(BM * BN / 4);
        if (!loader)
#pragma unroll
            for (int c = 0; c < NV4; ++c) {
                float4 sum = make_float4(0.f, 0.f, 0.f, 0.f);
                const float4* src = base + c * 512 + threadIdx.x;
                // eight slabs in flight, added in slice order; a last group of fewer is padded with zeros (x + 0 = x: the sum's bits do
                // not depend on the padding) -- one slab per round trip, the first form of this tail, cost ~2 us per slice: a launch in
                // 6 slices took longer than the same launch in 8
                for (int sl = 0; sl < g.split; sl += 8) {
                    float4 v[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u)
                        v[u] = sl + u < g.split ? src[(size_t)(sl + u) * (BM * BN / 4)] : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                    for (int u = 0; u < 8; ++u) { sum.x += v[u].x; sum.y += v[u].y; sum.z += v[u].z; sum.w += v[u].w; }
                }
                const int i = c / (TN * 4), j = (c / 4) % TN, r4 = c % 4;
                acc[0][i][j][4 * r4] = sum.x; acc[0][i][j][4 * r4 + 1] = sum.y; acc[0][i][j][4 * r4 + 2] = sum.z; acc[0][i][j][4 * r4 + 3] = sum.w;
            }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __syncthreads();                                        // the ring becomes the epilogue's wave-private scratch
    if (loader) return;

    // ---- epilogue.  C/D layout of the 32x32 MFMA: col = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5).
    auto& av = acc[0];
    unsigned char* const ws = smem + wid * Q::EPI;
    const long mw = m0 + wm * WM, nw0 = n0 + wn * WN;       // first row / column of this wave's tile
    constexpr int SROW = Q::SROW, TROW = Q::TROW;
    const bool edge = mw + WM > g.M || nw0 + WN > g.N;      // (wave-uniform)
    const int mrem = (int)min((long)WM, g.M - mw) - 4 * lh; // rows ml = c + 4 lh of the wave's tile are valid while c < mrem
    auto rowc = [](int i, int r) { return i * 32 + (r & 3) + 8 * (r >> 2); };
    const int mlim_ = edge ? mrem : (1 << 20);
    auto row_limit = [&]() { int v = mlim_; asm volatile("" : "+v"(v)); return v; };

    if constexpr (NP == 2) {
        if (g.metaOut && tile == 0 && threadIdx.x == 0) g.metaOut[0] = out_scale;
        if (g.metaZero && tile == 0 && threadIdx.x < META_SLOTS) g.metaZero[2 + threadIdx.x] = 0.f;
    }
    // final values in place of the accumulators
    const float slope = g.act == 1 ? 0.f : (g.act == 2 ? 0.01f : 1.f);
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const long n = nw0 + j * 32 + l31;
        const float bv = (g.bias != nullptr && n < g.N) ? g.bias[n] : 0.f;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float v = alpha * av[i][j][r] + bv;
                av[i][j][r] = v > 0.f ? v : v * slope;
            }
    }
    if (g.mask_mode) {
        // piece 0 of the wave's WM x 32 mask block j: 64 B per row -> LDS image [m][64 B + 16]; OOB rows read as zero = masked out
        const __amdgpu_buffer_rsrc_t rsM = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(g.maskx), 0, (int)g.mask_bytes, 0x00020000);
        constexpr int MROW = 80;
        const float mslope = g.mask_mode == 2 ? 0.01f : 0.f;   // 2: LeakyReLU'(0.01) of the masked activation (VPoser's encoder)
        const int row0 = lane >> 2, ch = lane & 3;          // 16 rows per load
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const long nb = nw0 + j * 32;
            const bool nok = nb < g.N;
            i32x4 mv[WM / 16];
#pragma unroll
            for (int u = 0; u < WM / 16; ++u) {
                const int off = nok ? (int)(((mw + row0 + 16 * u) * g.ldmask + (nb >> 5) * (32 * NP)) * 2 + ch * 16) : OOB;
                mv[u] = __builtin_bit_cast(i32x4, __builtin_amdgcn_raw_buffer_load_b128(rsM, off, 0, 0));
            }
#pragma unroll
            for (int u = 0; u < WM / 16; ++u) *reinterpret_cast<i32x4*>(ws + (row0 + 16 * u) * MROW + ch * 16) = mv[u];
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const short mb = *reinterpret_cast<const short*>(ws + (rowc(i, r) + 4 * lh) * MROW + l31 * 2);
                    av[i][j][r] = mb > 0 ? av[i][j][r] : av[i][j][r] * mslope;   // bf16 / fp16 > 0  <=>  its bits, as int16, > 0
                }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
    }
    if (edge) {                                             // pads of the xp copies stay zero; column sums skip them
        const int mlim = row_limit();
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int lim = nw0 + j * 32 + l31 >= g.N ? -1 : mlim;
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) av[i][j][r] = rowc(i, r) >= lim ? 0.f : av[i][j][r];
        }
    }

    if constexpr (NP == 2) {
        if (g.metaOut) {                                    // the result's true absmax (rows / columns beyond the matrix: zero unless
            float mx = 0.f;                                 // an interior tile, where every element is valid)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int r = 0; r < 16; ++r) mx = fmaxf(mx, fabsf(av[i][j][r]));
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off, 64));
            if (lane == 0) meta_absmax_put(g.metaOut, tile * 8 + wid, mx);
        }
    }

    if (g.colsum) {
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const long n = nw0 + j * 32 + l31;
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                float csum = 0.f;
#pragma unroll
                for (int r = 0; r < 16; ++r) csum += av[i][j][r];
                csum += __shfl_xor(csum, 32, 64);          // lanes l and l + 32: the same column, the other rows of the band
                const long band = (mw + i * 32) / 32;
                if (lh == 0 && n < g.N && band < 2 * ((g.M + 63) / 64)) g.colsum[band * g.ldcs + n] = csum;
            }
        }
    }

    if (g.C) {
        const __amdgpu_buffer_rsrc_t rsC = __builtin_amdgcn_make_buffer_rsrc(g.C, 0, (int)(unsigned)(((g.M - 1) * g.ldc + g.N) * 4), 0x00020000);
        const int ldc4 = (int)(g.ldc * 4);
        const int so1 = ldc4, so2 = 2 * ldc4, so3 = 3 * ldc4;
        auto store_c = [&](auto addc) {
            const int mlim = row_limit();
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const long n = nw0 + j * 32 + l31;
                const int off0 = n < g.N ? (int)(((mw + 4 * lh) * g.ldc + n) * 4) : OOB;
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int base = n < g.N ? off0 + (i * 32 + 8 * q) * ldc4 : OOB;
                        int off[4];
                        float v[4];
#pragma unroll
                        for (int c = 0; c < 4; ++c) {
                            off[c] = i * 32 + 8 * q + c >= mlim ? OOB : base;
                            v[c] = av[i][j][4 * q + c];
                        }
                        if constexpr (decltype(addc)::value) {
                            float old[4];
                            old[0] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsC, off[0], 0, 0));
                            old[1] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsC, off[1], so1, 0));
                            old[2] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsC, off[2], so2, 0));
                            old[3] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsC, off[3], so3, 0));
#pragma unroll
                            for (int c = 0; c < 4; ++c) v[c] += old[c];
                        }
                        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v[0]), rsC, off[0], 0, 0);
                        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v[1]), rsC, off[1], so1, 0);
                        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v[2]), rsC, off[2], so2, 0);
                        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v[3]), rsC, off[3], so3, 0);
                    }
            }
        };
        if (g.out_mode) store_c(std::true_type{}); else store_c(std::false_type{});
    }

    if (g.Cx) {
        // per 32-column block j (= one k-block of the copy): [m][piece][32 n] image by element writes, then whole KBB-byte row
        // segments out, 16 B per lane (plain global stores under a lane predicate: gemm_b16x.h on buffer stores with soffset)
        const int mvalid = row_limit() + 4 * lh;            // valid rows m of the wave's tile
#pragma unroll
        for (int j = 0; j < TN; ++j) {
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    unsigned short pc[NP];
                    split_pieces<NP>(av[i][j][r] * out_scale, pc);
#pragma unroll
                    for (int p = 0; p < NP; ++p)
                        *reinterpret_cast<unsigned short*>(ws + (rowc(i, r) + 4 * lh) * SROW + p * 64 + l31 * 2) = pc[p];
                }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            const long nb = nw0 + j * 32;
            unsigned short* dst = g.Cx + mw * g.ldcx + (nb >> 5) * (32 * NP);
#pragma unroll
            for (int p = 0; p < CH; ++p) {                  // WM * CH chunks, 64 per pass
                const int idx = p * 64 + lane, row = idx / CH, ch = idx - row * CH;
                const i32x4 v = *reinterpret_cast<const i32x4*>(ws + row * SROW + ch * 16);
                if (nb < g.N && row < mvalid) *reinterpret_cast<i32x4*>(dst + (long)row * g.ldcx + ch * 8) = v;
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
    }

    if (g.CxT) {
        // per 32-column block j: [n 32][k-block i][piece][32 m] image -- a lane's registers r = 4 q .. 4 q + 3 are four CONSECUTIVE
        // rows m -> one 8-byte write per piece; then TM * KBB-byte row segments out, 16 B per lane
        const long mblocks = (g.M + 31) / 32;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    unsigned short pc[4][NP];
#pragma unroll
                    for (int c = 0; c < 4; ++c) split_pieces<NP>(av[i][j][4 * q + c] * out_scale, pc[c]);
#pragma unroll
                    for (int p = 0; p < NP; ++p) {
                        uint2 pk;
                        pk.x = (unsigned)pc[0][p] | ((unsigned)pc[1][p] << 16);
                        pk.y = (unsigned)pc[2][p] | ((unsigned)pc[3][p] << 16);
                        *reinterpret_cast<uint2*>(ws + l31 * TROW + i * KBB + p * 64 + (8 * q + 4 * lh) * 2) = pk;
                    }
                }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            const long nb = nw0 + j * 32;
            constexpr int CHR = TM * CH;                    // chunks per image row
#pragma unroll
            for (int p = 0; p < CHR / 2; ++p) {             // 32 * CHR chunks, 64 per pass
                const int idx = p * 64 + lane, row = idx / CHR, ch = idx - row * CHR;
                const i32x4 v = *reinterpret_cast<const i32x4*>(ws + row * TROW + ch * 16);
                const long mb = (mw >> 5) + ch / CH;        // the k-block of the copy this chunk belongs to
                if (nb + row < g.N && mb < mblocks)
                    *reinterpret_cast<i32x4*>(g.CxT + (nb + row) * g.ldcxt + (mw >> 5) * (32 * NP) + ch * 8) = v;
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
    }
}

template <int NP, int BN>
__global__ __launch_bounds__(768) void gemm_xp_kernel(Args g) { gemm_xp_body<NP, BN>(g, (int)blockIdx.x, (int)gridDim.x); }

// Several independent products in ONE launch (round 6: the four parameter gradients dW_l = dY_l^T X_l of the MotionNet backward, behind
// the dX chain instead of beside it -- every launch of this kernel takes whole CUs, so dX and dW launches side by side only time-share).
// Problem i owns blocks [first[i], first[i] + nwg[i]); first[i] is a multiple of 8 so that a problem's block b still runs on XCD b % 8;
// the blocks in between exit at once.
constexpr int MAX_GROUP = 4;
struct GroupArgs { Args p[MAX_GROUP]; int first[MAX_GROUP]; int nwg[MAX_GROUP]; int n; };

template <int NP, int BN>
__global__ __launch_bounds__(768) void gemm_xp_grouped_kernel(GroupArgs a) {
    int i = 0;
#pragma unroll
    for (int q = 1; q < MAX_GROUP; ++q)
        if (q < a.n && (int)blockIdx.x >= a.first[q]) i = q;
    const int bid = (int)blockIdx.x - a.first[i];
    if (bid >= a.nwg[i]) return;                            // (block-uniform: padding between problems)
    gemm_xp_body<NP, BN>(a.p[i], bid, a.nwg[i]);
}

template <int NP, int BN>
hipError_t launch_grouped(const GroupArgs& a, int blocks, hipStream_t s) {
    using Q = Geo<NP, BN>;
    static NemoAttrOnce attr_once;
    auto kern = &gemm_xp_grouped_kernel<NP, BN>;
    if (attr_once.need()) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, Q::LDS);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(Q::THREADS), Q::LDS, s, a);
    return hipSuccess;
}

template <int NP, int BN>
hipError_t launch(const Args& g, hipStream_t s) {
    using Q = Geo<NP, BN>;
    static NemoAttrOnce attr_once;
    auto kern = &gemm_xp_kernel<NP, BN>;
    if (attr_once.need()) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, Q::LDS);
        if (e != hipSuccess) return e;
    }
    const long blocks = (long)g.tiles_m * g.tiles_n * (g.split > 1 ? g.split : 1);
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(Q::THREADS), Q::LDS, s, g);
    return hipSuccess;
}

inline long ld_for(int np, long k) { return 32L * np * ((k + 31) / 32); }

// Fills tiles / slices / extents; false when the problem does not fit the kernel's addressing (32-bit buffer offsets).
inline bool plan(Args& g, int np, int bn, int split) {
    g.tiles_m = (int)((g.M + 127) / 128);
    g.tiles_n = (int)((g.N + bn - 1) / bn);
    if (split < 1) split = 1;
    long kc = (g.K + split - 1) / split;
    kc = (kc + 31) / 32 * 32;
    g.k_chunk = kc;
    g.split = (int)((g.K + kc - 1) / kc);
    const long kw = ld_for(np, g.K);
    if (g.lda < kw || g.ldb < kw || (g.lda & 7) || (g.ldb & 7) || g.K < 1) return false;
    const long a_el = (g.M - 1) * g.lda + kw, b_el = (g.N - 1) * g.ldb + kw;
    // every offset the kernel forms stays below 2^31 (OOB = 2^31 is then beyond every extent, with or without the scalar part)
    // (soffset = tile row * ld + k-block, voffset = row in tile (< 128 + bn) * ld + chunk)
    if ((g.tiles_m * 128L + 128) * g.lda * 2 + kw * 2 >= (1L << 31) || ((long)g.tiles_n * bn + bn) * g.ldb * 2 + kw * 2 >= (1L << 31)) return false;
    if (g.C && (g.M + 512) * g.ldc * 4 >= (1L << 31)) return false;
    g.a_bytes = (unsigned)(a_el * 2);
    g.b_bytes = (unsigned)(b_el * 2);
    g.mask_bytes = 0;
    if (g.mask_mode) {
        const long nw = ld_for(np, g.N);
        if (g.ldmask < nw || (g.M + 512) * g.ldmask * 2 >= (1L << 31)) return false;
        g.mask_bytes = (unsigned)(((g.M - 1) * g.ldmask + nw) * 2);
    }
    if (g.Cx && g.ldcx < ld_for(np, g.N)) return false;
    if (g.CxT && g.ldcxt < ld_for(np, g.M)) return false;
    return true;
}

// ---- fp32 -> xp copies (plain [r][c] and / or transposed [c][r]) of a row-major matrix: 32 x 32 tiles through LDS, every k-block of
// the destinations written whole (pads zero).  One launch converts a LIST of matrices (the weights of the chain in one launch).
struct CastDesc {
    const float* src; long rows, cols, lds;
    unsigned short* dst; long ldd;                          // [rows][xp over cols] or NULL
    unsigned short* dstT; long lddT;                        // [cols][xp over rows] or NULL
    float scale;
    float* meta;                                            // fmt 2, may be NULL: scale record, absmax in, scale = 2^floor(log2(2^15 / absmax)) out
    int tile0, tiles_c;                                     // first block of this matrix in the launch, 32-column tiles per row of tiles
};
struct CastArgs { CastDesc d[MAX_CAST]; int n; };

template <int NP>
__global__ __launch_bounds__(256) void cast_xp_kernel(CastArgs a) {
    __shared__ float t[32][33];
    int di = 0;
#pragma unroll
    for (int i = 1; i < MAX_CAST; ++i)
        if (i < a.n && (int)blockIdx.x >= a.d[i].tile0) di = i;
    const CastDesc& d = a.d[di];
    const int bt = (int)blockIdx.x - d.tile0;
    const long r0 = (long)(bt / d.tiles_c) * 32, c0 = (long)(bt % d.tiles_c) * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    float scale = d.scale;
    if (NP == 2 && d.meta) {
        scale = pow2_scale(meta_absmax(d.meta), 15);
        if (bt == 0 && threadIdx.x == 0) d.meta[0] = scale;
    }
    if (d.rows <= 0 || d.cols <= 0) return;                   // (block-uniform)
    {
        // the thread's four loads are UNCONDITIONAL, from indices clamped into the matrix, and in flight together: behind the bounds
        // predicate hipcc compiled the loop to one load + wait per iteration (four dependent round trips per block)
        float v[4];
        const long cc = min(c0 + tx, d.cols - 1);
#pragma unroll
        for (int q = 0; q < 4; ++q) v[q] = d.src[min(r0 + ty + 8 * q, d.rows - 1) * d.lds + cc];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int i = ty + 8 * q;
            t[i][tx] = (r0 + i < d.rows && c0 + tx < d.cols) ? v[q] * scale : 0.f;
        }
    }
    __syncthreads();
    if (d.dst) {
        for (int i = ty; i < 32; i += 8) {
            const long r = r0 + i;
            if (r >= d.rows) continue;
            unsigned short pc[NP];
            split_pieces<NP>(t[i][tx], pc);
            unsigned short* o = d.dst + r * d.ldd + (c0 >> 5) * (32 * NP) + tx;
#pragma unroll
            for (int p = 0; p < NP; ++p) o[p * 32] = pc[p];
        }
    }
    if (d.dstT) {
        for (int i = ty; i < 32; i += 8) {
            const long c = c0 + i;                          // destination row = source column
            if (c >= d.cols) continue;
            unsigned short pc[NP];
            split_pieces<NP>(t[tx][i], pc);
            unsigned short* o = d.dstT + c * d.lddT + (r0 >> 5) * (32 * NP) + tx;
#pragma unroll
            for (int p = 0; p < NP; ++p) o[p * 32] = pc[p];
        }
    }
}

__global__ __launch_bounds__(256) void absmax_kernel(AbsmaxArgs a) { absmax_block(a, (int)blockIdx.x, (int)gridDim.x); }

}  // namespace xp
