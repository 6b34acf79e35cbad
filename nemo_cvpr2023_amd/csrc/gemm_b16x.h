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

constexpr int BK = 64;                                      // bf16 per K tile: a tile row is 128 B = 8 chunks of 16 B

template <int BM, int BN, int WGM, int WGN>
struct Geo {
    static constexpr int NW = WGM * WGN;
    static_assert(NW == 8, "8 waves");
    static constexpr int WM = BM / WGM, WN = BN / WGN, TM = WM / 32, TN = WN / 32;
    static_assert(WM % 32 == 0 && WN % 32 == 0, "wave tile in 32 x 32 accumulators");
    static constexpr int PA = BM / 8 / NW, PB = BN / 8 / NW, G = PA + PB;       // DMA pieces per wave and K tile
    static_assert(BM % 64 == 0 && BN % 64 == 0, "whole pieces per wave");
    static constexpr int STAGE = (BM + BN) * 128;                                 // bytes
    // epilogue scratch per wave: the larger of the [m][n] image (row stride WN * 2 + 16 B) and the [n][m] image
    // (row stride WM * 2 + 16 B)
    static constexpr int SROW = WN * 2 + 16, TROW = WM * 2 + 16;
    static constexpr int EPI = (WM * SROW > WN * TROW ? WM * SROW : WN * TROW);
    static constexpr int LDS = (2 * STAGE > NW * EPI ? 2 * STAGE : NW * EPI);
};

// keep the first `nvalid` (<= 8, may be <= 0) bf16 of an operand fragment
__device__ __forceinline__ bf16x8 keep_first(bf16x8 v, int nvalid) {
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    u32x4 u = __builtin_bit_cast(u32x4, v);
#pragma unroll
    for (int d = 0; d < 4; ++d) {
        const int e = nvalid - 2 * d;
        u[d] = e >= 2 ? u[d] : (e == 1 ? (u[d] & 0xffffu) : 0u);
    }
    return __builtin_bit_cast(bf16x8, u);
}

__device__ __forceinline__ unsigned short bf16_bits(float v) { return __builtin_bit_cast(unsigned short, (__bf16)v); }

template <int BM, int BN, int WGM, int WGN>
__global__ __launch_bounds__(512) void gemm_b16x_kernel(Args g) {
    using Q = Geo<BM, BN, WGM, WGN>;
    constexpr int WM = Q::WM, WN = Q::WN, TM = Q::TM, TN = Q::TN, PA = Q::PA, G = Q::G;
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
    const long klen = kend > kbeg ? kend - kbeg : 0;
    const int nt = (int)((klen + BK - 1) / BK);

    const int lane = threadIdx.x & 63;
    const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = wid / WGN, wn = wid % WGN;
    const int l31 = lane & 31, lh = lane >> 5;

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // ---- LDS-DMA addressing.  Piece u of an operand tile = rows 8 u .. 8 u + 7; this wave issues pieces wid, wid + 8, ...
    // ((u & 1) == (wid & 1): the piece-dependent swizzle term is a per-wave constant, one lane offset serves them all).
    const unsigned smem_byte = (unsigned)reinterpret_cast<unsigned long long>(smem);
    auto rsrc = [](const void* p, unsigned bytes) {
        const unsigned long long b = reinterpret_cast<unsigned long long>(p);
        return i32x4{(int)(unsigned)b, (int)(unsigned)((b >> 32) & 0xffffu), (int)bytes, 0x00020000};
    };
    const i32x4 rsA = rsrc(g.A, g.a_bytes), rsB = rsrc(g.B, g.b_bytes);
    const int dr = lane >> 3;
    const int dchunk = (lane & 7) ^ ((dr >> 1) | (4 * (wid & 1)));
    const int voffA = (int)((dr * g.lda + 8 * dchunk) * 2), voffB = (int)((dr * g.ldb + 8 * dchunk) * 2);
    const unsigned rowA = (unsigned)(((m0 + 8 * wid) * g.lda + kbeg) * 2), rowB = (unsigned)(((n0 + 8 * wid) * g.ldb + kbeg) * 2);
    const unsigned stepA = (unsigned)(64 * g.lda * 2), stepB = (unsigned)(64 * g.ldb * 2);
    // piece p (0 .. G - 1) of this wave for K tile t into stage (t & 1)
    auto dma = [&](int p, int t) {
        const unsigned st = smem_byte + (unsigned)((t & 1) * Q::STAGE);
        if (p < PA) glds::dma_piece(rsA, st + (unsigned)((wid + 8 * p) * 1024), voffA, rowA + (unsigned)p * stepA + (unsigned)t * (BK * 2));
        else glds::dma_piece(rsB, st + (unsigned)(BM * 128 + (wid + 8 * (p - PA)) * 1024), voffB,
                             rowB + (unsigned)(p - PA) * stepB + (unsigned)t * (BK * 2));
    };

    // ---- operand fetch: row (in the wave's 32-row block) l31, chunk 2 s + lh of k-step s; (row >> 1) & 7 == (l31 >> 1) & 7
    // because every 32-row block starts at a multiple of 32
    const int sw = (l31 >> 1) & 7;
    const unsigned fa0 = (unsigned)((wm * WM + l31) * 128), fb0 = (unsigned)(BM * 128 + (wn * WN + l31) * 128);
    auto compute = [&](int t, auto tailc, auto issuec) {
        constexpr bool TAIL = decltype(tailc)::value, ISSUE = decltype(issuec)::value;
        const unsigned char* st = smem + (t & 1) * Q::STAGE;
        const int krem = (int)(klen - (long)t * BK);            // (TAIL: < 64)
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            bf16x8 fa[TM], fb[TN];
            const unsigned co = (unsigned)(((2 * s + lh) ^ sw) << 4);
#pragma unroll
            for (int i = 0; i < TM; ++i) fa[i] = *reinterpret_cast<const bf16x8*>(st + fa0 + i * 32 * 128 + co);
#pragma unroll
            for (int j = 0; j < TN; ++j) fb[j] = *reinterpret_cast<const bf16x8*>(st + fb0 + j * 32 * 128 + co);
            if constexpr (TAIL) {
                const int nv = krem - (16 * s + 8 * lh);
#pragma unroll
                for (int i = 0; i < TM; ++i) fa[i] = keep_first(fa[i], nv);
#pragma unroll
                for (int j = 0; j < TN; ++j) fb[j] = keep_first(fb[j], nv);
            }
            if constexpr (ISSUE) {
                // two pieces of tile t + 1 per k-step, front-loaded: the last piece has at least a quarter of the tile's
                // MFMAs plus the barrier to land
#pragma unroll
                for (int p = 2 * s; p < 2 * s + 2; ++p)
                    if (p < G) dma(p, t + 1);
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);
        }
    };

    // ---- K loop (nt >= 1: every slice is non-empty by construction).  Tile t lives in stage t & 1.  Iteration t: this
    // wave's pieces of tile t have landed (vmcnt), barrier (everyone's have; everyone is done reading tile t - 1), tile t + 1
    // is requested into the stage tile t - 1 occupied while tile t is multiplied.
    // The loop body exists ONCE (tile t + 1 requested, no masking); the last tile runs behind the loop through the masking
    // variant whether it is partial or not (a few dozen VALU operations once per block) -- with several variants selected
    // inside the loop hipcc copies all accumulators between two register sets on every iteration.
#pragma unroll
    for (int p = 0; p < G; ++p) dma(p, 0);
    for (int t = 0; t < nt - 1; ++t) {
        glds::wait_vmcnt<0>();
        asm volatile("" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        compute(t, std::false_type{}, std::true_type{});
    }
    glds::wait_vmcnt<0>();
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    compute(nt - 1, std::true_type{}, std::false_type{});

    // ---- split-K: publish the partial tile write-through, take a ticket; the last arriver sums all slices in slice order
    // (deterministic) and runs the epilogue.  Slab layout = the register image: float4 #(i, j, r4) of thread t.
    constexpr int NV4 = TM * TN * 4;
    if (g.split > 1) {
        float4* slab = reinterpret_cast<float4*>(g.slabs) + ((size_t)tile * g.split + slice) * (size_t)(BM * BN / 4);
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
        __syncthreads();                                    // (also: every wave is done reading the stages)
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
#pragma unroll
        for (int c = 0; c < NV4; ++c) {
            float4 sum = make_float4(0.f, 0.f, 0.f, 0.f);
            const float4* src = base + c * 512 + threadIdx.x;
            int sl = 0;
            for (; sl + 8 <= g.split; sl += 8) {            // eight slabs in flight, added in slice order
                float4 v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) v[u] = src[(size_t)(sl + u) * (BM * BN / 4)];
#pragma unroll
                for (int u = 0; u < 8; ++u) { sum.x += v[u].x; sum.y += v[u].y; sum.z += v[u].z; sum.w += v[u].w; }
            }
            for (; sl < g.split; ++sl) {
                const float4 v = src[(size_t)sl * (BM * BN / 4)];
                sum.x += v.x; sum.y += v.y; sum.z += v.z; sum.w += v.w;
            }
            const int i = c / (TN * 4), j = (c / 4) % TN, r4 = c % 4;
            acc[i][j][4 * r4] = sum.x; acc[i][j][4 * r4 + 1] = sum.y; acc[i][j][4 * r4 + 2] = sum.z; acc[i][j][4 * r4 + 3] = sum.w;
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __syncthreads();                                        // the stages become the epilogue's wave-private scratch

    // ---- epilogue.  C/D layout of the 32x32 MFMA: col = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5).
    unsigned char* const ws = smem + wid * Q::EPI;
    const long mw = m0 + wm * WM, nw0 = n0 + wn * WN;       // first row / column of this wave's tile
    constexpr int SROW = Q::SROW, TROW = Q::TROW;

    if (g.mask_mode) {
        // the wave's WM x WN mask tile as whole 128-byte row segments (16 B per lane, OOB rows read as zero = masked out,
        // they are never stored) -> LDS image [m][n]
        const __amdgpu_buffer_rsrc_t rsM = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(g.mask16), 0, (int)g.mask_bytes, 0x00020000);
        constexpr int CH = WN / 8;                          // 16-byte chunks per row
        constexpr int NP = WM * CH / 64;
        i32x4 mv[NP];
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            const int idx = p * 64 + lane, row = idx / CH, ch = idx % CH;
            const long off = ((mw + row) * g.ldmask16 + nw0 + 8 * ch) * 2;
            mv[p] = __builtin_bit_cast(i32x4, __builtin_amdgcn_raw_buffer_load_b128(rsM, (int)off, 0, 0));
        }
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            const int idx = p * 64 + lane, row = idx / CH, ch = idx % CH;
            *reinterpret_cast<i32x4*>(ws + row * SROW + ch * 16) = mv[p];
        }
    }

    // final values in place of the accumulators (acc[i][j][r] := v)
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const long n = nw0 + j * 32 + l31;
        const float bv = (g.bias != nullptr && n < g.N) ? g.bias[n] : 0.f;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            float csum = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int ml = i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;        // row inside the wave's tile
                float v = g.alpha * acc[i][j][r] + bv;
                if (g.act == 1) v = v > 0.f ? v : 0.f;
                else if (g.act == 2) v = v > 0.f ? v : 0.01f * v;
                if (g.mask_mode) {
                    const unsigned short mb = *reinterpret_cast<const unsigned short*>(ws + ml * SROW + (j * 32 + l31) * 2);
                    v = ((mb & 0x7fffu) != 0 && !(mb & 0x8000u)) ? v : 0.f;     // bf16 > 0
                }
                if (mw + ml >= g.M || n >= g.N) v = 0.f;                        // pads of the bf16 copies stay zero
                acc[i][j][r] = v;
                csum += v;
            }
            if (g.colsum) {
                csum += __shfl_xor(csum, 32, 64);          // lanes l and l + 32: the same column, the other rows of the band
                const long band = (mw + i * 32) / 32;
                if (lh == 0 && n < g.N && band < 2 * ((g.M + 63) / 64)) g.colsum[band * g.ldcs + n] = csum;
            }
            if (g.C) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const long m = mw + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                    if (m < g.M && n < g.N) {
                        float* c = g.C + m * g.ldc + n;
                        if (g.out_mode == 0) *c = acc[i][j][r];
                        else *c += acc[i][j][r];
                    }
                }
            }
        }
    }

    if (g.Cb) {
        // [m][n] image: element writes, then whole 128-byte row segments out, 16 B per lane
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // (the mask reads above are done: same region)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int ml = i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                    *reinterpret_cast<unsigned short*>(ws + ml * SROW + (j * 32 + l31) * 2) = bf16_bits(acc[i][j][r]);
                }
        constexpr int CH = WN / 8, NP = WM * CH / 64;
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            const int idx = p * 64 + lane, row = idx / CH, ch = idx % CH;
            const i32x4 v = *reinterpret_cast<const i32x4*>(ws + row * SROW + ch * 16);
            const long m = mw + row, n = nw0 + 8 * ch;
            if (m < g.M && n < g.N) *reinterpret_cast<i32x4*>(g.Cb + m * g.ldcb + n) = v;     // (n + 8 <= ldcb: host-checked)
        }
    }

    if (g.CbT) {
        // [n][m] image: a lane's registers r = 4 q .. 4 q + 3 are four CONSECUTIVE rows m -> one 8-byte write; then whole
        // 2 WM-byte row segments out, 16 B per lane
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    uint2 pk;
                    pk.x = (unsigned)bf16_bits(acc[i][j][4 * q]) | ((unsigned)bf16_bits(acc[i][j][4 * q + 1]) << 16);
                    pk.y = (unsigned)bf16_bits(acc[i][j][4 * q + 2]) | ((unsigned)bf16_bits(acc[i][j][4 * q + 3]) << 16);
                    *reinterpret_cast<uint2*>(ws + (j * 32 + l31) * TROW + (i * 32 + 8 * q + 4 * lh) * 2) = pk;
                }
        constexpr int CH = WM / 8, NP = (WN * CH + 63) / 64;
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            const int idx = p * 64 + lane, row = idx / CH, ch = idx % CH;
            if (WN * CH % 64 != 0 && row >= WN) break;
            const i32x4 v = *reinterpret_cast<const i32x4*>(ws + row * TROW + ch * 16);
            const long n = nw0 + row, m = mw + 8 * ch;
            if (n < g.N && m < g.M) *reinterpret_cast<i32x4*>(g.CbT + n * g.ldcbt + m) = v;   // (m + 8 <= ldcbt: host-checked)
        }
    }
}

template <int BM, int BN, int WGM, int WGN>
hipError_t launch(const Args& g, hipStream_t s) {
    using Q = Geo<BM, BN, WGM, WGN>;
    static bool attr_set = false;
    auto kern = &gemm_b16x_kernel<BM, BN, WGM, WGN>;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, Q::LDS);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    const long blocks = (long)g.tiles_m * g.tiles_n * (g.split > 1 ? g.split : 1);
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(512), Q::LDS, s, g);
    return hipSuccess;
}

// tile configurations: 0 = 192 x 256, 1 = 256 x 256, 2 = 128 x 256 (all 2 x 4 waves)
inline int tile_bm(int cfg) { return cfg == 0 ? 192 : cfg == 1 ? 256 : 128; }
inline hipError_t launch_cfg(int cfg, const Args& g, hipStream_t s) {
    if (cfg == 0) return launch<192, 256, 2, 4>(g, s);
    if (cfg == 1) return launch<256, 256, 2, 4>(g, s);
    return launch<128, 256, 2, 4>(g, s);
}

// Fills tiles / slices / extents; false when the problem does not fit the kernel's addressing (32-bit buffer offsets).
inline bool plan(Args& g, int cfg, int split) {
    const int BM = tile_bm(cfg), BN = 256;
    g.tiles_m = (int)((g.M + BM - 1) / BM);
    g.tiles_n = (int)((g.N + BN - 1) / BN);
    if (split < 1) split = 1;
    long kc = (g.K + split - 1) / split;
    kc = (kc + BK - 1) / BK * BK;
    g.k_chunk = kc;
    g.split = (int)((g.K + kc - 1) / kc);
    auto up8 = [](long x) { return (x + 7) / 8 * 8; };
    const long a_el = (g.M - 1) * g.lda + (up8(g.K) < g.lda ? up8(g.K) : g.lda);
    const long b_el = (g.N - 1) * g.ldb + (up8(g.K) < g.ldb ? up8(g.K) : g.ldb);
    const long a_max = (g.M + 512) * g.lda + g.K + 128, b_max = (g.N + 512) * g.ldb + g.K + 128;
    if (a_max * 2 >= (1L << 32) || b_max * 2 >= (1L << 32) || g.K < 1) return false;
    g.a_bytes = (unsigned)(a_el * 2);
    g.b_bytes = (unsigned)(b_el * 2);
    g.mask_bytes = 0;
    if (g.mask_mode) {
        const long m_el = (g.M - 1) * g.ldmask16 + (up8(g.N) < g.ldmask16 ? up8(g.N) : g.ldmask16);
        if ((g.M + 512) * g.ldmask16 * 2 >= (1L << 32)) return false;
        g.mask_bytes = (unsigned)(m_el * 2);
    }
    return true;
}

}  // namespace b16x
