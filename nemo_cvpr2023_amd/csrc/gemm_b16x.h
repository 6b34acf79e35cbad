// Large-tile bf16 GEMM for the bf16-in-memory MLP chain (round 5; BASELINE configs[2], the roofline configuration):
//   C (M x N) (op)= epilogue(alpha * A B^T),  A (M x K) and B (N x K) bf16 IN MEMORY, both k-contiguous.
// Replaces, under args.gemm_dtype = 'bf16', the hidden-layer nn.Linear products of MotionNet and their autograd
// (nemo/neural_motion_model.py:58-71, :130-148; human_body_prior/models/vposer_model.py:68-106): forward Y = X W^T,
// activation gradient dX = dY W (B = the transposed weight copy) and parameter gradient dW = dY^T X (both operands the
// transposed activation copies, K = samples, cut into K slices across workgroups).
//
// Why a kernel of its own (not a parameter of gemm_glds.h): that kernel's 64 x 64 x 64 bf16 tile moves 16 KiB through
// the CU's LDS-DMA path for 128 matrix-pipe cycles per wave -- it is bound by DMA issue at 0.13 of the bf16 peak
// (profiles/r04_kernel_trace_c3b.md).  Here a workgroup is 8 waves (2 per SIMD) on a BM x BN tile of 192 x 256 /
// 256 x 256 / 128 x 256; a wave owns WM x WN = 96 x 64 (128 x 64, 64 x 64) as TM x TN accumulators of
// v_mfma_f32_32x32x16_bf16, so ONE ds_read_b128 per operand row feeds TN (TM) MFMAs and a K tile of 64 costs
// (BM + BN) x 128 B of LDS-DMA for TM x TN x 4 MFMAs per wave: 56 KiB per 1536 matrix-pipe cycles at 192 x 256.
//
// Operand tiles travel global -> LDS by LDS-DMA (buffer_load_dwordx4 ... lds; 1 KiB pieces = 8 rows x 128 B), two LDS
// stages, one s_barrier per K tile; the pieces of tile t + 1 are issued two at a time between the MFMA groups of tile t.
// LDS image of a tile: [row][8 chunks of 16 B], chunk c of row r at slot c ^ ((r >> 1) & 7) -- the swizzle sits on the
// SOURCE address of the DMA (a piece lands lane-linear); the 16 lanes a ds_read_b128 services together then touch 16
// distinct 16-byte bank quads (same image as gemm_glds.h's "image K").
//
// Bounds: both operands are read through buffer descriptors ending at the operand's last valid byte (rows beyond M / N
// return zeros).  The end of K inside a K tile is masked PER ELEMENT in registers (both operands), so nothing is assumed
// about what sits behind column K of a row (a transposed activation copy is only zeroed up to the next even column).
//
// Epilogue (fused; what the chain needs): v = maskfn(act(alpha * acc + bias)); any of
//   C   fp32 (store / +=),
//   Cb  bf16 copy   [m][n]  -- through a wave-private LDS transposition: whole 128-byte row segments, 16 B per lane,
//   CbT bf16 copy^T [n][m]  -- likewise: 2 WM-byte row segments, 16 B per lane,
//   colsum: per 32-row band column sums of v (one writer per element: deterministic) = the layer's bias gradient.
// The activation mask (ReLU': mask16[m][n] > 0) is read as whole lines into LDS and picked up per element from there.
// Split-K (dW): write-through slabs + ticket, the last arriver sums the slices in slice order (deterministic).
#pragma once
#include <type_traits>
#include "common.h"
#include "gemm_glds.h"

namespace b16x {

using glds::bf16x8;
using glds::f32x16;
using glds::i32x4;
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

struct Args {
    const unsigned short* A; const unsigned short* B;       // bf16 bit patterns
    long M, N, K;                                           // K in bf16 elements
    long lda, ldb;                                          // elements; multiples of 8, bases 16-byte aligned
    float* C; long ldc; int out_mode;                       // C may be NULL; 0 store, 1 +=
    const float* bias; int act; float alpha;                // act 0 / 1 ReLU / 2 LeakyReLU(0.01)
    const unsigned short* mask16; long ldmask16; int mask_mode;   // 0 none, 1: v = mask > 0 ? v : 0
    unsigned short* Cb; long ldcb;                          // ldcb % 8 == 0, ldcb >= N rounded up to 8
    unsigned short* CbT; long ldcbt;                        // ldcbt % 8 == 0, ldcbt >= M rounded up to 8
    float* colsum; long ldcs;                               // rows: one per 32-row band, 2 ceil(M / 64) of them
    float* slabs; int* counters;
    long k_chunk; int split;                                // K range per slice (multiple of 64)
    int tiles_m, tiles_n;
    unsigned a_bytes, b_bytes, mask_bytes;                  // buffer extents
};

// BK (template parameter BKT): bf16 per ring stage.  32: a stage row is 64 B = 4 chunks of 16 B, pieces of 16 rows (round 5, first
// form).  64: a stage row is a whole 128-byte line = 8 chunks, pieces of 8 rows -- with every CU streaming, 64-byte row segments
// reach HALF the L2 -> LDS rate of whole lines (profiles/r05_dma_rate.txt: 13 - 21 against 25 - 45 GB/s per CU), and the K loop
// of the 192 x 256 tile needs 87 GB/s per CU at the full MFMA rate: the 32-wide form sits on that limit (~700 TFLOP/s).

// LW: loader waves (0: the eight MFMA waves request their operand tiles themselves; 4: four more waves -- one per SIMD --
// issue every LDS-DMA piece and the MFMA waves never stall on a DMA issue, ~100 - 180 cycles each beside MFMAs and LDS
// reads: with eight pieces per wave and 64 k that was as long as the MFMAs themselves)
template <int BM, int BN, int WGM, int WGN, int NST, int LW = 0, int BKT = 32>
struct Geo {
    static_assert(BKT == 32 || BKT == 64, "ring stage of 32 or 64 k");
    static constexpr int BK = BKT, ROWB = 2 * BKT, NCH = ROWB / 16, PR = 1024 / ROWB, KS = BKT / 16;   // row bytes, chunks, rows per piece, k-steps
    static constexpr int NW = WGM * WGN;
    static_assert(NW == 8, "8 MFMA waves");
    static_assert(LW == 0 || LW == 4, "loader waves: none or one per SIMD");
    static constexpr int THREADS = 64 * (NW + LW);
    static constexpr int WM = BM / WGM, WN = BN / WGN, TM = WM / 32, TN = WN / 32;
    static_assert(WM % 32 == 0 && WN % 32 == 0, "wave tile in 32 x 32 accumulators");
    static constexpr int NPA = BM / PR, NPB = BN / PR;                            // 1-KiB DMA pieces (PR rows x ROWB bytes) per stage
    static_assert(NPA % 2 == 0, "the swizzle of a piece depends on its parity only");
    static constexpr int NI = LW ? LW : NW;                                       // waves that issue DMAs
    static constexpr int GW = (NPA + NPB + NI - 1) / NI;                          // pieces per issuing wave and stage (waves whose
                                                                                  // last piece does not exist issue a dummy: same counts)
    static constexpr int STAGE = (BM + BN) * ROWB;                                // bytes
    static constexpr int RING = NST * STAGE;
    // epilogue scratch per wave: the larger of the [m][n] image (row stride WN * 2 + 16 B) and the [n][m] image
    // (row stride WM * 2 + 16 B)
    static constexpr int SROW = WN * 2 + 16, TROW = WM * 2 + 16;
    static constexpr int EPI = (WM * SROW > WN * TROW ? WM * SROW : WN * TROW);
    static constexpr int LDS = (RING + 1024 > NW * EPI ? RING + 1024 : NW * EPI);  // (+ 1 KiB: where dummy pieces land)
};

// keep the first `nvalid` (<= 8, may be <= 0) bf16 of an operand fragment
__device__ __forceinline__ bf16x8 keep_first(bf16x8 v, int nvalid) {
    u32x4 u = __builtin_bit_cast(u32x4, v);
#pragma unroll
    for (int d = 0; d < 4; ++d) {
        const int e = nvalid - 2 * d;
        u[d] = e >= 2 ? u[d] : (e == 1 ? (u[d] & 0xffffu) : 0u);
    }
    return __builtin_bit_cast(bf16x8, u);
}

__device__ __forceinline__ unsigned short bf16_bits(float v) { return __builtin_bit_cast(unsigned short, (__bf16)v); }
__device__ __forceinline__ unsigned pack_bf16(float lo, float hi) { return (unsigned)bf16_bits(lo) | ((unsigned)bf16_bits(hi) << 16); }

constexpr int OOB = (int)0x80000000u;      // a buffer offset beyond every descriptor of this kernel (extents < 2^31)

template <int BM, int BN, int WGM, int WGN, int NST, int LW, int BKT>
__global__ __launch_bounds__(64 * (8 + LW)) void gemm_b16x_kernel(Args g) {
    using Q = Geo<BM, BN, WGM, WGN, NST, LW, BKT>;
    constexpr int WM = Q::WM, WN = Q::WN, TM = Q::TM, TN = Q::TN, GW = Q::GW, NPA = Q::NPA, NPB = Q::NPB, NI = Q::NI;
    constexpr int BK = Q::BK, ROWB = Q::ROWB, NCH = Q::NCH, PR = Q::PR, KS = Q::KS;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];          // the ONLY LDS object of the kernel

    // ---- block -> (K slice, row tile, column tile).  Block b runs on XCD b % 8 (observed; speed only): give every XCD a
    // CONTIGUOUS run of the (slice, tm, tn) order with tn fastest, so the blocks of an XCD share A row panels in its L2.
    const int nwg = (int)gridDim.x, bid = (int)blockIdx.x;
    const int xq = nwg >> 3, xr = nwg & 7, xcd = bid & 7;
    const int lin = (xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq) + (bid >> 3);
    const int tn = lin % g.tiles_n, tm = (lin / g.tiles_n) % g.tiles_m, slice = lin / (g.tiles_n * g.tiles_m);
    const int tile = tn * g.tiles_m + tm;
    const long m0 = (long)tm * BM, n0 = (long)tn * BN;
    const long kbeg = (long)slice * g.k_chunk;
    const long kend = g.split > 1 ? min(g.K, kbeg + g.k_chunk) : g.K;
    const int klen = (int)(kend > kbeg ? kend - kbeg : 0);
    const int nt = (klen + BK - 1) / BK;                    // >= 1: every slice is non-empty by construction

    const int lane = threadIdx.x & 63;
    const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    __builtin_assume(wid >= 0 && wid < 8 + LW);
    const bool loader = LW > 0 && wid >= 8;                 // (wave-uniform)
    const int iw = LW > 0 ? wid - 8 : wid;                  // index among the issuing waves (loaders: 0 .. LW - 1)
    const int wm = wid / WGN, wn = wid % WGN;
    const int l31 = lane & 31, lh = lane >> 5;

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // ---- LDS-DMA addressing.  Stage image: [row][4 chunks of 16 B], chunk c of row r at slot c ^ ((r >> 2) & 3): the 16
    // lanes a ds_read_b128 services together (rows {0-3, 12-15, 20-27} + 4 a of one chunk index) then touch 16 distinct
    // 16-byte bank quads.  A piece = 16 rows; a piece's lane (row dr = lane >> 2, slot lane & 3) fetches source chunk
    // (lane & 3) ^ ((dr >> 2) & 3) -- the swizzle does not depend on the piece.  Issuing wave w of NI issues pieces w, w + NI,
    // ... of the NPA + NPB pieces of a stage (A's first).
    const unsigned smem_byte = (unsigned)reinterpret_cast<unsigned long long>(smem);
    auto rsrc = [](const void* p, unsigned bytes) {
        const unsigned long long b = reinterpret_cast<unsigned long long>(p);
        return i32x4{(int)(unsigned)b, (int)(unsigned)((b >> 32) & 0xffffu), (int)bytes, 0x00020000};
    };
    const i32x4 rsA = rsrc(g.A, g.a_bytes), rsB = rsrc(g.B, g.b_bytes);
    // (BK = 64: chunk c of row r at slot c ^ ((r >> 1) & 7), gemm_glds.h's "image K"; a piece holds 8 rows, so the swizzle of
    //  its lane depends on the piece's parity: (dr >> 1) | 4 (piece & 1))
    const int dr = lane / NCH, dslot = lane % NCH;
    auto voff_of = [&](long ld, int par) {
        const int swz = BK == 32 ? ((dr >> 2) & 3) : ((dr >> 1) | (par << 2));
        return (int)((dr * ld + 8 * (dslot ^ swz)) * 2);
    };
    const unsigned baseA = (unsigned)((m0 * g.lda + kbeg) * 2), baseB = (unsigned)((n0 * g.ldb + kbeg) * 2);
    const unsigned rowsA = (unsigned)(PR * g.lda * 2), rowsB = (unsigned)(PR * g.ldb * 2);
    // the p-th piece (0 .. GW - 1) of this wave: which operand, where in a stage, from where -- selected once, branch-free
    // (q = iw + NI p).  Pieces that do not exist (q >= NPA + NPB) and tiles beyond the slice are requested out of bounds
    // (nothing is fetched; zeros land in the 1 KiB behind the ring resp. in a stage nobody reads any more): every wave has the
    // same number of DMAs in flight per tile, which is what the counted vmcnt below relies on.
    i32x4 prs[GW];
    unsigned plds[GW], psrc[GW];
    int pvoff[GW];
    bool pdummy[GW];
#pragma unroll
    for (int p = 0; p < GW; ++p) {
        const int q = iw + NI * p;
        const bool isA = q < NPA;
        pdummy[p] = q >= NPA + NPB;
        prs[p] = isA ? rsA : rsB;
        plds[p] = pdummy[p] ? (unsigned)Q::RING : (isA ? (unsigned)(q * 1024) : (unsigned)(BM * ROWB + (q - NPA) * 1024));
        psrc[p] = isA ? baseA + (unsigned)q * rowsA : baseB + (unsigned)(q - NPA) * rowsB;
        pvoff[p] = pdummy[p] ? OOB : voff_of(isA ? g.lda : g.ldb, q & 1);     // (NPA is even: q and q - NPA have one parity)
    }
    // piece p of K tile t into ring stage `stage`
    auto dma = [&](int p, int t, int stage) {
        const unsigned st = pdummy[p] ? 0u : (unsigned)(stage * Q::STAGE);
        glds::dma_piece(prs[p], smem_byte + st + plds[p], t < nt ? pvoff[p] : OOB, psrc[p] + (unsigned)t * ROWB);
    };

    // ---- operand fetch.  k-step s (0 / 1) of a stage, lane half lh: chunk 2 s + lh of row l31 of each 32-row block;
    // (row >> 2) & 3 == (l31 >> 2) & 3 because every 32-row block starts at a multiple of 32.
    struct Frag { bf16x8 a[TM], b[TN]; };
    const int sw = BK == 32 ? ((l31 >> 2) & 3) : ((l31 >> 1) & 7);
    const unsigned fa0 = (unsigned)((wm * WM + l31) * ROWB), fb0 = (unsigned)(BM * ROWB + (wn * WN + l31) * ROWB);
    auto fetch = [&](Frag& f, int stage, int s) {
        const unsigned char* st = smem + stage * Q::STAGE;
        const unsigned co = (unsigned)(((2 * s + lh) ^ sw) << 4);
#pragma unroll
        for (int i = 0; i < TM; ++i) f.a[i] = *reinterpret_cast<const bf16x8*>(st + fa0 + i * 32 * ROWB + co);
#pragma unroll
        for (int j = 0; j < TN; ++j) f.b[j] = *reinterpret_cast<const bf16x8*>(st + fb0 + j * 32 * ROWB + co);
    };
    auto mask_tail = [&](Frag& f, int t, int s) {           // elements at or beyond the end of K
        const int nv = klen - t * BK - (16 * s + 8 * lh);
#pragma unroll
        for (int i = 0; i < TM; ++i) f.a[i] = keep_first(f.a[i], nv);
#pragma unroll
        for (int j = 0; j < TN; ++j) f.b[j] = keep_first(f.b[j], nv);
    };
    // the MFMAs of one k-step.  `after_first()` runs behind the FIRST of them: the next fragment reads are issued there, so
    // that the wait hipcc puts in front of this k-step's first MFMA finds nothing younger outstanding (its scoreboard is
    // conservative next to the inline-asm DMAs: a wait for these operands placed behind younger reads becomes lgkmcnt(0)
    // and would stall on those).  `issue_t` >= 0: this wave's pieces of that tile leave one at a time behind the MFMAs
    // that follow.
    auto mma = [&](const Frag& f, auto&& after_first, int issue_t, int stage) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.a[i], f.b[j], acc[i][j], 0, 0, 0);
                if (i == 0 && j == 0) {
                    __builtin_amdgcn_sched_barrier(0);
                    after_first();
                    __builtin_amdgcn_sched_barrier(0);
                }
                if (LW == 0 && issue_t >= 0 && i * TN + j >= 1 && i * TN + j - 1 < GW) {
                    dma(i * TN + j - 1, issue_t, stage);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
    };
    static_assert(LW > 0 || GW < TM * TN, "one piece behind each of GW MFMAs");

    // ---- K loop over a ring of NST stages (tile t in stage t % NST), software-pipelined by one k-step ACROSS the barrier:
    //   iteration t:  MFMAs of F = (t, k-step 0), G = fragments (t, k-step 1) read behind the first of them;
    //                 tile t + 1 has landed (counted vmcnt: NST - 2 younger tiles stay in flight), G has arrived (lgkmcnt),
    //                 s_barrier: tile t + 1 is visible to every wave, nobody reads stage t % NST any more;
    //                 MFMAs of G, F = fragments (t + 1, k-step 0) read behind the first of them, tile t + NST requested
    //                 into stage t % NST piece by piece behind the next ones.
    // After every barrier a wave has MFMAs whose operands are already in registers while its next reads are in flight.
    // The loop body exists once; the last tile runs behind it through the masking variant whether it is partial or not
    // (with several variants selected inside the loop hipcc copies all accumulators between two register sets per iteration).
    // With loader waves (LW > 0) the same schedule is split by role: a loader's iteration is { tile t + 1 has landed (its
    // pieces: counted vmcnt); s_barrier; request tile t + NST into stage t % NST }, an MFMA wave's is the one above without
    // the waits for and the requests of DMAs; both pass the same barriers.
    Frag F, G;
    if (loader) {
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
        if (LW == 0) {
#pragma unroll
            for (int t = 0; t < NST; ++t)
#pragma unroll
                for (int p = 0; p < GW; ++p) dma(p, t, t);
            glds::wait_vmcnt<(NST - 1) * GW>();
        }
        asm volatile("" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        fetch(F, 0, 0);
        int cur = 0;                                        // stage of tile t
        // (KS k-steps per tile, fragments alternating F, G, ...; the LAST k-step of a tile sits behind the barrier)
        for (int t = 0; t < nt - 1; ++t) {
#pragma unroll
            for (int s = 0; s + 1 < KS; ++s) {
                if (s & 1) mma(G, [&] { fetch(F, cur, s + 1); }, -1, 0);
                else mma(F, [&] { fetch(G, cur, s + 1); }, -1, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
            if (LW == 0) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"((NST - 2) * GW) : "memory");
            else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            const int nxt = cur + 1 == NST ? 0 : cur + 1;
            mma(G, [&] { fetch(F, nxt, 0); }, t + NST, cur);      // (KS is even: the last k-step's fragments are G's)
            __builtin_amdgcn_sched_barrier(0);
            cur = nxt;
        }
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            Frag& f = (s & 1) ? G : F;
            if (s > 0) fetch(f, cur, s);
            mask_tail(f, nt - 1, s);
            mma(f, [] {}, -1, 0);
        }
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");     // (dummy pieces of the tiles behind the slice)
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
                    const glds::f32x4 vv = {acc[i][j][4 * r4], acc[i][j][4 * r4 + 1], acc[i][j][4 * r4 + 2], acc[i][j][4 * r4 + 3]};
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
            // eight slabs in flight, added in slice order; a last group of fewer padded with zeros (x + 0 = x; round 6: the tail used to
            // take one round trip per slab)
            for (int sl = 0; sl < g.split; sl += 8) {
                float4 v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u)
                    v[u] = sl + u < g.split ? src[(size_t)(sl + u) * (BM * BN / 4)] : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                for (int u = 0; u < 8; ++u) { sum.x += v[u].x; sum.y += v[u].y; sum.z += v[u].z; sum.w += v[u].w; }
            }
            const int i = c / (TN * 4), j = (c / 4) % TN, r4 = c % 4;
            acc[i][j][4 * r4] = sum.x; acc[i][j][4 * r4 + 1] = sum.y; acc[i][j][4 * r4 + 2] = sum.z; acc[i][j][4 * r4 + 3] = sum.w;
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __syncthreads();                                        // the ring becomes the epilogue's wave-private scratch
    if (loader) return;

    // ---- epilogue.  C/D layout of the 32x32 MFMA: col = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5).  Everything
    // per element is branch-free; what depends on the call (mask, edge tiles, which outputs) is decided once per pass.
    unsigned char* const ws = smem + wid * Q::EPI;
    const long mw = m0 + wm * WM, nw0 = n0 + wn * WN;       // first row / column of this wave's tile
    constexpr int SROW = Q::SROW, TROW = Q::TROW;
    const bool edge = mw + WM > g.M || nw0 + WN > g.N;      // (wave-uniform)
    const int mrem = (int)min((long)WM, g.M - mw) - 4 * lh; // rows ml = c + 4 lh of the wave's tile are valid while c < mrem
    auto rowc = [](int i, int r) { return i * 32 + (r & 3) + 8 * (r >> 2); };
    // one row limit for every pass (interior tiles: no row is beyond it), re-materialised per pass: shared between the
    // passes, hipcc keeps ~50 compare masks alive in scalar registers and spills them
    const int mlim_ = edge ? mrem : (1 << 20);
    auto row_limit = [&]() { int v = mlim_; asm volatile("" : "+v"(v)); return v; };

    if (g.mask_mode) {
        // the wave's WM x WN mask tile as whole 128-byte row segments (16 B per lane; OOB rows read as zero = masked out,
        // they are never stored) -> LDS image [m][n]; four loads in flight
        const __amdgpu_buffer_rsrc_t rsM = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(g.mask16), 0, (int)g.mask_bytes, 0x00020000);
        constexpr int CH = WN / 8, NP = WM * CH / 64;
        static_assert(NP % 4 == 0, "mask tile in batches of four loads");
        const int row0 = lane / CH, ch = lane % CH;         // (64 / CH rows per load)
        const int moff = (int)(((mw + row0) * g.ldmask16 + nw0 + 8 * ch) * 2);
        const int mstep = (int)((64 / CH) * g.ldmask16 * 2);
#pragma unroll
        for (int p0 = 0; p0 < NP; p0 += 4) {
            i32x4 mv[4];
#pragma unroll
            for (int u = 0; u < 4; ++u)
                mv[u] = __builtin_bit_cast(i32x4, __builtin_amdgcn_raw_buffer_load_b128(rsM, moff, (p0 + u) * mstep, 0));
#pragma unroll
            for (int u = 0; u < 4; ++u) *reinterpret_cast<i32x4*>(ws + (row0 + (p0 + u) * (64 / CH)) * SROW + ch * 16) = mv[u];
        }
        asm volatile("" ::: "memory");              // (the image is written and read through different types: no reordering across)
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
                const float v = g.alpha * acc[i][j][r] + bv;
                acc[i][j][r] = v > 0.f ? v : v * slope;
            }
    }
    if (g.mask_mode) {
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const short mb = *reinterpret_cast<const short*>(ws + (rowc(i, r) + 4 * lh) * SROW + (j * 32 + l31) * 2);
                    acc[i][j][r] = mb > 0 ? acc[i][j][r] : 0.f;                // bf16 > 0  <=>  its bits, as int16, > 0
                }
        asm volatile("" ::: "memory");              // (the image is written and read through different types: no reordering across)
    }
    if (edge) {                                             // pads of the bf16 copies stay zero; column sums skip them
        const int mlim = row_limit();
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int lim = nw0 + j * 32 + l31 >= g.N ? -1 : mlim;
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = rowc(i, r) >= lim ? 0.f : acc[i][j][r];
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
                for (int r = 0; r < 16; ++r) csum += acc[i][j][r];
                csum += __shfl_xor(csum, 32, 64);          // lanes l and l + 32: the same column, the other rows of the band
                const long band = (mw + i * 32) / 32;
                if (lh == 0 && n < g.N && band < 2 * ((g.M + 63) / 64)) g.colsum[band * g.ldcs + n] = csum;
            }
        }
    }

    if (g.C) {
        // through a buffer descriptor: a lane's 32-bit offset, rows / columns beyond the matrix pointed out of bounds
        const __amdgpu_buffer_rsrc_t rsC = __builtin_amdgcn_make_buffer_rsrc(g.C, 0, (int)(unsigned)(((g.M - 1) * g.ldc + g.N) * 4), 0x00020000);
        const int ldc4 = (int)(g.ldc * 4);
        // a lane's four consecutive rows (r & 3) go through the scalar offset (4 values), the row group (i, r >> 2) through the
        // lane offset (one add each): few scalar registers (one scalar offset per element used to spill ~100 of them)
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
                            v[c] = acc[i][j][4 * q + c];
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

    if (g.Cb) {
        // [m][n] image: element writes, then whole 128-byte row segments out, 16 B per lane
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    *reinterpret_cast<unsigned short*>(ws + (rowc(i, r) + 4 * lh) * SROW + (j * 32 + l31) * 2) = bf16_bits(acc[i][j][r]);
        asm volatile("" ::: "memory");              // (the image is written and read through different types: no reordering across)
        // (plain global stores under a lane predicate, NOT buffer stores with a scalar offset: hipcc's hazard recognizer
        //  assumes a >64-bit MUBUF store with a REGISTER soffset has read its data registers at once and re-used the first of
        //  them in the next VALU instruction -- on gfx950 the store then wrote that scratch value; tools/gemm_b16x_dev check)
        constexpr int CH = WN / 8, NP = WM * CH / 64;
        const int row0 = lane / CH, ch = lane % CH;
        const int mvalid = row_limit() + 4 * lh;            // valid rows m of the wave's tile
        const bool nok = nw0 + 8 * ch < g.N;                // (n + 8 <= ldcb: host-checked)
        unsigned short* dst = g.Cb + (mw + row0) * g.ldcb + nw0 + 8 * ch;
        const long ostep = (64 / CH) * g.ldcb;
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            const i32x4 v = *reinterpret_cast<const i32x4*>(ws + (row0 + p * (64 / CH)) * SROW + ch * 16);
            if (nok && row0 + p * (64 / CH) < mvalid) *reinterpret_cast<i32x4*>(dst + p * ostep) = v;
        }
        asm volatile("" ::: "memory");              // (the image is written and read through different types: no reordering across)
    }

    if (g.CbT) {
        // [n][m] image: a lane's registers r = 4 q .. 4 q + 3 are four CONSECUTIVE rows m -> one 8-byte write; then whole
        // 2 WM-byte row segments out, 16 B per lane
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    uint2 pk;
                    pk.x = pack_bf16(acc[i][j][4 * q], acc[i][j][4 * q + 1]);
                    pk.y = pack_bf16(acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]);
                    *reinterpret_cast<uint2*>(ws + (j * 32 + l31) * TROW + (i * 32 + 8 * q + 4 * lh) * 2) = pk;
                }
        asm volatile("" ::: "memory");              // (the image is written and read through different types: no reordering across)
        constexpr int CH = WM / 8, NP = (WN * CH + 63) / 64;
        const int mvalid = row_limit() + 4 * lh;            // valid rows m of the wave's tile
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            const int idx = p * 64 + lane, row = idx / CH, ch = idx % CH;
            if (WN * CH % 64 != 0 && row >= WN) break;
            const i32x4 v = *reinterpret_cast<const i32x4*>(ws + row * TROW + ch * 16);
            if (nw0 + row < g.N && 8 * ch < mvalid)         // (m + 8 <= ldcbt: host-checked)
                *reinterpret_cast<i32x4*>(g.CbT + (nw0 + row) * g.ldcbt + mw + 8 * ch) = v;
        }
    }
}

template <int BM, int BN, int WGM, int WGN, int NST, int LW, int BKT = 32>
hipError_t launch(const Args& g, hipStream_t s) {
    using Q = Geo<BM, BN, WGM, WGN, NST, LW, BKT>;
    static NemoAttrOnce attr_once;
    auto kern = &gemm_b16x_kernel<BM, BN, WGM, WGN, NST, LW, BKT>;
    if (attr_once.need()) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, Q::LDS);
        if (e != hipSuccess) return e;
    }
    const long blocks = (long)g.tiles_m * g.tiles_n * (g.split > 1 ? g.split : 1);
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(Q::THREADS), Q::LDS, s, g);
    return hipSuccess;
}

// tile configurations: 0 = 192 x 256, 1 = 256 x 256, 2 = 128 x 256 (all 2 x 4 MFMA waves), + 3: the same with four loader
// waves (192 x 256 and 128 x 256: their register budgets allow three waves per SIMD); ring depth: 5 stages where they fit
#ifndef B16X_NST
#define B16X_NST 5
#endif
constexpr int NST_DEFAULT = B16X_NST;
inline int tile_bm(int cfg) { return cfg % 3 == 0 ? 192 : cfg % 3 == 1 ? 256 : 128; }
inline hipError_t launch_cfg(int cfg, const Args& g, hipStream_t s) {
    if (cfg == 0) return launch<192, 256, 2, 4, NST_DEFAULT, 0>(g, s);
    if (cfg == 1) return launch<256, 256, 2, 4, 4, 0>(g, s);
    if (cfg == 2) return launch<128, 256, 2, 4, NST_DEFAULT, 0>(g, s);
    if (cfg == 3) return launch<192, 256, 2, 4, NST_DEFAULT, 4>(g, s);
    if (cfg == 5) return launch<128, 256, 2, 4, NST_DEFAULT, 4>(g, s);
    // whole-line ring stages (BK = 64): 6 = 192 x 256 + 4L, two stages (112 KiB); 8 = 128 x 256 + 4L, three (144 KiB)
    if (cfg == 6) return launch<192, 256, 2, 4, 2, 4, 64>(g, s);
    if (cfg == 8) return launch<128, 256, 2, 4, 3, 4, 64>(g, s);
    return hipErrorInvalidValue;
}

// Fills tiles / slices / extents; false when the problem does not fit the kernel's addressing (32-bit buffer offsets).
inline bool plan(Args& g, int cfg, int split) {
    if (cfg == 4 || cfg == 7 || cfg < 0 || cfg > 8) return false;
    const int BM = tile_bm(cfg), BN = 256;
    g.tiles_m = (int)((g.M + BM - 1) / BM);
    g.tiles_n = (int)((g.N + BN - 1) / BN);
    if (split < 1) split = 1;
    long kc = (g.K + split - 1) / split;
    kc = (kc + 63) / 64 * 64;
    g.k_chunk = kc;
    g.split = (int)((g.K + kc - 1) / kc);
    auto up8 = [](long x) { return (x + 7) / 8 * 8; };
    const long a_el = (g.M - 1) * g.lda + (up8(g.K) < g.lda ? up8(g.K) : g.lda);
    const long b_el = (g.N - 1) * g.ldb + (up8(g.K) < g.ldb ? up8(g.K) : g.ldb);
    const long a_max = (g.M + 512) * g.lda + g.K + 128, b_max = (g.N + 512) * g.ldb + g.K + 128;
    // every offset the kernel forms stays below 2^31 (OOB = 2^31 is then beyond every extent, with or without the scalar part)
    if (a_max * 2 >= (1L << 31) || b_max * 2 >= (1L << 31) || g.K < 1) return false;
    if (g.C && (g.M + 512) * g.ldc * 4 >= (1L << 31)) return false;
    if (g.Cb && (g.M + 512) * g.ldcb * 2 >= (1L << 31)) return false;
    if (g.CbT && (g.N + 512) * g.ldcbt * 2 >= (1L << 31)) return false;
    g.a_bytes = (unsigned)(a_el * 2);
    g.b_bytes = (unsigned)(b_el * 2);
    g.mask_bytes = 0;
    if (g.mask_mode) {
        const long m_el = (g.M - 1) * g.ldmask16 + (up8(g.N) < g.ldmask16 ? up8(g.N) : g.ldmask16);
        if ((g.M + 512) * g.ldmask16 * 2 >= (1L << 31)) return false;
        g.mask_bytes = (unsigned)(m_el * 2);
    }
    return true;
}

}  // namespace b16x
