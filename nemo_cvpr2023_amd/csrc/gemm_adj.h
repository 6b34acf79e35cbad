// The blend-shape adjoint  dPF (M x 207) (+)= dVP^T P^T  (human_body_prior/body_model/lbs.py:229-233 backward; K = 3 NV =
// 20 670) on ONE 64 x 208 column tile per workgroup with MIXED MFMA shapes (round 4).
//
// The 64 x 64 plan of gemm_glds.h covers 207 columns with four 64-wide tiles: 24 % of its MFMAs multiply padding and the
// (K x M) operand dVP^T is streamed once per column tile (4.6 x its size in fetch traffic).  Round 2's single 64 x 208
// tile on v_mfma_f32_16x16x4_f32 read dVP^T once but paid the narrow instruction's doubled LDS operand reads on EVERY
// column and tied.  Here columns [0, 192) run on v_mfma_f32_32x32x2_f32 (wave w: rows [32 (w & 1), +32) x three 32-column
// blocks starting at 96 (w >> 1)) and only the 16-column remainder [192, 208) on v_mfma_f32_16x16x4_f32 (wave w: rows
// [16 w, +16)): 48 + 8 MFMAs per wave and K tile = 3328 pipe cycles for 851 968 FLOP -- the pipe's full rate -- with
// 1.3 % row padding and 0.5 % column padding instead of 24 %.
//
// Operand tiles through LDS-DMA exactly as in gemm_glds.h (same Operand class, same images and k permutation): A = dVP^T
// is image M (row-contiguous, 64 rows: 8 pieces per K tile), B = the blend shapes image K (k-contiguous, 208 rows: 26
// pieces -- row 207 reads zeros through the buffer descriptor).  dVP^T is read exactly once, straight from HBM (the mesh
// kernel has just written its 198 MB), the blend shapes 38 times from L2 / the Infinity Cache: the A ring is THREE K tiles
// deep (8 KiB each), the B ring two (26 KiB each) -- 76 KiB, two workgroups per CU.  Per iteration the pieces of B (t + 1)
// and then of A (t + 2) leave one at a time between the first MFMAs of tile t (an LDS-DMA issue holds the wave's
// instruction stream ~100 cycles); the wait at the top of an iteration leaves the two youngest pieces -- A's -- in flight.
// K is cut into slices across workgroups (write-through slabs + ticket, combined by the last arriver in slice order:
// deterministic), as in gemm_glds.h.
#pragma once
#include "gemm_glds.h"

namespace glds {

constexpr int ADJ_BM = 64, ADJ_BN = 208;
constexpr int ADJ_B_FLOATS = ADJ_BN * BK, ADJ_NA = 3, ADJ_NB = 2;
constexpr int adj_lds_bytes(int nwv) { return (ADJ_NA * 16 * nwv * BK + ADJ_NB * ADJ_B_FLOATS) * 4; }
constexpr int ADJ_LDS_BYTES = adj_lds_bytes(4);

// B16 (round 4, last third): the same tile for the bf16-in-memory chain (BASELINE configs[2]) -- dPF (+)= dVP P^T with dVP
// (samples x K) and P (207 x K) bf16 IN MEMORY, both k-contiguous (nemo_gemm_bf16mem's layout; A / B / lda / ldb / K arrive as
// the fp32-typed view of the same bytes, i.e. in units of bf16 pairs).  A K tile is 64 bf16 = the same 128-byte rows: A is
// image K too (8 pieces per tile), columns [0, 192) on v_mfma_f32_32x32x16_bf16 (4 k-steps x 3 per wave and K tile), the
// 16-column remainder on v_mfma_f32_16x16x32_bf16 (2 per wave and K tile).  The 64 x 64 plan read the 339 MB of d vp four
// times (once per column tile: 1.36 GB per 8192-sample launch at 5.4 TB/s -- bandwidth-bound) and spent 24 % of its MFMAs on
// padding.
// B16 = 2 (round 5, last third): the operands are fp16 PIECES in memory (two planes each: x0 = fp16(s x), x1 = fp16(s x - x0)) and the
// product is the fp32-equivalent sum A0 B0^T + A0 B1^T + A1 B0^T: the K slices of a row tile are dealt over the three
// (plane of A, plane of B) pairs (Args::nseg / seg_a / seg_b) and summed by the same ordered slab combine; v_mfma_f32_32x32x16_f16 /
// 16x16x32_f16 on the same images.  nemo_blend_adjoint_split.
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
// NWV = 8 (round 5, the split-precision form): a 128 x 208 tile on eight waves -- the blend-shape tiles (26 of the 34 KB a K tile of
// the 64-row form moves) are fetched once per 128 rows: the kernel is bound by the L2 -> LDS stream, not by the matrix pipe.
template <int B16, int NWV = 4>
__device__ __forceinline__ void gemm_adj_body(const Args& g, float* smem) {
    constexpr int ADJ_BM = 16 * NWV, ADJ_A_FLOATS = ADJ_BM * BK, NTH = 64 * NWV;
    using OA = Operand<ADJ_BM, B16 != 0, NWV>;
    using OB = Operand<ADJ_BN, true, NWV>;
    constexpr int GA = OA::PER_WAVE, GB = OB::NI / NWV + 1;      // 2 pieces of A per wave and K tile, 6 or 7 (3 or 4) of B
    const int bid = (int)blockIdx.x;
    const int tile = bid % g.tiles_m, slice = bid / g.tiles_m, split = g.split;
    const long m0 = (long)tile * ADJ_BM;
    const int spp = split / g.nseg, seg = slice / spp;           // slices per (plane, plane) pair; this slice's pair
    const long kbeg = (long)(slice - seg * spp) * g.k_chunk;
    const long kend = min(g.K, kbeg + g.k_chunk);
    const long klen = kend > kbeg ? kend - kbeg : 0;
    const int nt = (int)((klen + BK - 1) / BK);
    const bool tail = (klen % BK) != 0;                          // (B is image K: masked fill of a partial last tile)
    const int nfull = tail ? nt - 1 : nt;

    const int lane = threadIdx.x & 63;
    const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lr = lane & 31, lh = lane >> 5, lr16 = lane & 15, lh16 = lane >> 4;
    const int row32 = 32 * (wid % (NWV / 2)) + lr, col32 = 96 * (wid / (NWV / 2)) + lr;        // (+ 32 j)
    const int row16 = 16 * wid + lr16, col16 = 192 + lr16;

    f32x16 acc[3];
    f32x4 acc16 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;

    const unsigned smem_byte = (unsigned)reinterpret_cast<unsigned long long>(smem);
    OA oa; OB ob;
    const float* const Aq = g.A + g.seg_a[seg];
    const float* const Bq = g.B + g.seg_b[seg];
    oa.init(Aq, g.a_bytes, g.lda, lane, wid);
    ob.init(Bq, g.b_bytes, g.ldb, lane, wid);

    float* const sa = smem;                                       // A ring, then B ring
    float* const sb = smem + ADJ_NA * ADJ_A_FLOATS;
    const unsigned sa_byte = smem_byte, sb_byte = smem_byte + ADJ_NA * ADJ_A_FLOATS * 4;
    const bool b_extra = wid < OB::NI % NWV;                      // this wave issues one more piece of B

    // MFMAs of K tile t (A slot t % 3, B slot t % 2); meanwhile request B of tile tb and then A of tile ta (< 0: none)
    auto compute = [&](const float* as, const float* bs, int tb, int ta) {
        const unsigned nbb = sb_byte + (unsigned)(((tb < 0 ? 0 : tb) % ADJ_NB) * ADJ_B_FLOATS * 4);
        const unsigned nab = sa_byte + (unsigned)(((ta < 0 ? 0 : ta) % ADJ_NA) * ADJ_A_FLOATS * 4);
        const long nkb = kbeg + (long)(tb < 0 ? 0 : tb) * BK, nka = kbeg + (long)(ta < 0 ? 0 : ta) * BK;
        if constexpr (B16 != 0) {
            int piece = 0;
            auto issue_one = [&]() {
                if (piece < GB) {
                    if (tb >= 0 && (piece < GB - 1 || b_extra)) ob.dma_one(nbb, 0, nkb, wid, piece);
                } else if (piece < GB + GA) {
                    if (ta >= 0) oa.dma_one(nab, m0, nka, wid, piece - GB);
                }
                ++piece;
            };
            // the 16 x 16 x 32 operands: lane (row / column lane & 15, k group lane >> 4) holds 8 consecutive k = chunk 4 s2 + kq
            auto chunk16 = [](const float* tile, int row, int c) {
                return *reinterpret_cast<const bf16x8*>(tile + row * BK + ((c ^ ((row >> 1) & 7)) << 2));
            };
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const bf16x8 fa = OA::fetch16(as, row32, ks, lh);
                bf16x8 fb[3];
#pragma unroll
                for (int j = 0; j < 3; ++j) fb[j] = OB::fetch16(bs, col32 + 32 * j, ks, lh);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    if constexpr (B16 == 2) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, fa), __builtin_bit_cast(f16x8, fb[j]), acc[j], 0, 0, 0);
                    else acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb[j], acc[j], 0, 0, 0);
                    issue_one();                       // one LDS-DMA piece behind every MFMA: B's 6 - 7 first, A's 2 last
                    __builtin_amdgcn_sched_barrier(0);
                }
                if (ks & 1) {
                    const int s2 = ks >> 1;
                    const bf16x8 ga = chunk16(as, row16, 4 * s2 + lh16), gb = chunk16(bs, col16, 4 * s2 + lh16);
                    if constexpr (B16 == 2) acc16 = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, ga), __builtin_bit_cast(f16x8, gb), acc16, 0, 0, 0);
                    else acc16 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ga, gb, acc16, 0, 0, 0);
                }
            }
            return;
        }
        float fa[2][4], fb[2][3][4];
        OA::template fetch<8>(as, row32, 0, lh, fa[0]);
#pragma unroll
        for (int j = 0; j < 3; ++j) OB::template fetch<8>(bs, col32 + 32 * j, 0, lh, fb[0][j]);
        int piece = 0;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            if (q + 1 < 4) {
                OA::template fetch<8>(as, row32, q + 1, lh, fa[(q + 1) & 1]);
#pragma unroll
                for (int j = 0; j < 3; ++j) OB::template fetch<8>(bs, col32 + 32 * j, q + 1, lh, fb[(q + 1) & 1][j]);
            }
            __builtin_amdgcn_sched_barrier(0);         // the next group's operand reads stay AHEAD of this group's MFMAs
#pragma unroll
            for (int s = 0; s < 4; ++s) {
#pragma unroll
                for (int j = 0; j < 3; ++j)
                    acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[q & 1][s], fb[q & 1][j][s], acc[j], 0, 0, 0);
                // one LDS-DMA piece behind every step of the first groups: B's 6 - 7 first, A's 2 last
                if (piece < GB) {
                    if (tb >= 0 && (piece < GB - 1 || b_extra)) ob.dma_one(nbb, 0, nkb, wid, piece);
                } else if (piece < GB + GA) {
                    if (ta >= 0) oa.dma_one(nab, m0, nka, wid, piece - GB);
                }
                ++piece;
                __builtin_amdgcn_sched_barrier(0);
            }
            if (q == 1 || q == 3) {                    // the 16-column remainder: k group (q >> 1) of the narrow shape
                float ga[4], gb[4];
                OA::template fetch<16>(as, row16, q >> 1, lh16, ga);
                OB::template fetch<16>(bs, col16, q >> 1, lh16, gb);
#pragma unroll
                for (int s = 0; s < 4; ++s) acc16 = __builtin_amdgcn_mfma_f32_16x16x4f32(ga[s], gb[s], acc16, 0, 0, 0);
            }
        }
    };

    if (nfull > 0) {
        oa.dma(sa_byte, m0, kbeg, wid);
        ob.dma(sb_byte, 0, kbeg, wid);
        if (nfull > 1) oa.dma(sa_byte + ADJ_A_FLOATS * 4, m0, kbeg + BK, wid);
    }
    for (int t = 0; t < nfull; ++t) {
        // tile t has landed (this wave's pieces): everything but the two youngest pieces, which are A (t + 1)'s
        if (t + 1 < nfull) wait_vmcnt<GA>(); else wait_vmcnt<0>();
        asm volatile("" ::: "memory");
        __builtin_amdgcn_s_barrier();                  // ... everybody's, and everybody is done reading tile t - 1
        asm volatile("" ::: "memory");
        compute(sa + (t % ADJ_NA) * ADJ_A_FLOATS, sb + (t % ADJ_NB) * ADJ_B_FLOATS, t + 1 < nfull ? t + 1 : -1,
                t + 2 < nfull ? t + 2 : -1);
    }
    if (tail) {
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        const long k0 = kbeg + (long)nfull * BK;
        if constexpr (B16 != 0) oa.fill_tail(sa, Aq, m0, k0, g.M, kend);   // image K: masked through registers
        else oa.dma(sa_byte, m0, k0, wid);                                 // image M: k rows beyond K read zeros
        ob.fill_tail(sb, Bq, 0, k0, g.N, kend);                            // image K: masked through registers
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        compute(sa, sb, -1, -1);
    }

    // ---- split-K: publish the partial tile (13 float4 per thread); the slices of a tile are summed in a fixed order
    // (deterministic) by last arrivers -- in ONE level up to 16 slices, in TWO levels beyond (groups of 8 slices: a group's
    // last arriver sums its 8 slabs into a group slab, the last group's arriver sums the group slabs): a small batch needs
    // 40 - 64 K slices to fill the machine, and one workgroup reading 64 slabs of 52 KiB back to back was a 30+ us tail
    if (split > 1) {
        constexpr size_t TILE4 = (size_t)(ADJ_BM * ADJ_BN / 4);
        const int gs = split <= 16 ? split : 8;                     // slices per group
        const int ngroups = (split + gs - 1) / gs;
        const int grp = slice / gs, gsz = min(gs, split - grp * gs);
        float4* const slabs4 = reinterpret_cast<float4*>(g.slabs);
        float4* const gslabs4 = slabs4 + (size_t)g.tiles_m * split * TILE4;           // group slabs behind the slice slabs
        int* const cnt2 = g.counters + tile;                                           // groups of this tile that are complete
        int* const cnt1 = g.counters + g.tiles_m + tile * ngroups + grp;               // slices of this group that have arrived
        int* flag = reinterpret_cast<int*>(smem);
        auto put = [&](float4* slab, int idx, f32x4 vv) {
            float4* dst = slab + idx * NTH + threadIdx.x;
            asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(dst), "v"(vv) : "memory");
        };
        auto put_all = [&](float4* slab) {
#pragma unroll
            for (int j = 0; j < 3; ++j)
#pragma unroll
                for (int r4 = 0; r4 < 4; ++r4)
                    put(slab, j * 4 + r4, f32x4{acc[j][4 * r4], acc[j][4 * r4 + 1], acc[j][4 * r4 + 2], acc[j][4 * r4 + 3]});
            put(slab, 12, acc16);
        };
        // sum `n` consecutive slabs starting at `base` into acc / acc16, in order, two slabs' loads in flight
        auto sum_slabs = [&](const float4* base, int n) {
            float4 sum[13];
#pragma unroll
            for (int i = 0; i < 13; ++i) sum[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            int sl = 0;
            for (; sl + 1 < n; sl += 2) {
                const float4* p0 = base + (size_t)sl * TILE4 + threadIdx.x;
                const float4* p1 = p0 + TILE4;
                float4 v0[13], v1[13];
#pragma unroll
                for (int i = 0; i < 13; ++i) { v0[i] = p0[i * NTH]; v1[i] = p1[i * NTH]; }
#pragma unroll
                for (int i = 0; i < 13; ++i) {
                    sum[i].x += v0[i].x; sum[i].y += v0[i].y; sum[i].z += v0[i].z; sum[i].w += v0[i].w;
                    sum[i].x += v1[i].x; sum[i].y += v1[i].y; sum[i].z += v1[i].z; sum[i].w += v1[i].w;
                }
            }
            if (sl < n) {
                const float4* p0 = base + (size_t)sl * TILE4 + threadIdx.x;
#pragma unroll
                for (int i = 0; i < 13; ++i) {
                    const float4 v = p0[i * NTH];
                    sum[i].x += v.x; sum[i].y += v.y; sum[i].z += v.z; sum[i].w += v.w;
                }
            }
#pragma unroll
            for (int j = 0; j < 3; ++j)
#pragma unroll
                for (int r4 = 0; r4 < 4; ++r4) {
                    const float4 v = sum[j * 4 + r4];
                    acc[j][4 * r4] = v.x; acc[j][4 * r4 + 1] = v.y; acc[j][4 * r4 + 2] = v.z; acc[j][4 * r4 + 3] = v.w;
                }
            acc16 = f32x4{sum[12].x, sum[12].y, sum[12].z, sum[12].w};
        };
        // arrive on `cnt` (after this workgroup's write-through stores have left); true in the last of `n` arrivers, which
        // then sees everybody's slabs and has returned the counter to zero
        auto last_of = [&](int* cnt, int n) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (threadIdx.x == 0) *flag = __hip_atomic_fetch_add(cnt, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __syncthreads();
            const bool last = *flag == n - 1;
            if (last && threadIdx.x == 0) {
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                __hip_atomic_store(cnt, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            __syncthreads();
            return last;
        };
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                  // (smem is reused for the hand-off flag)
        put_all(slabs4 + ((size_t)tile * split + slice) * TILE4);
        if (!last_of(cnt1, gsz)) return;
        sum_slabs(slabs4 + ((size_t)tile * split + (size_t)grp * gs) * TILE4, gsz);
        if (ngroups > 1) {
            put_all(gslabs4 + ((size_t)tile * ngroups + grp) * TILE4);
            if (!last_of(cnt2, ngroups)) return;
            sum_slabs(gslabs4 + (size_t)tile * ngroups * TILE4, ngroups);
        }
    }

    // ---- epilogue: C = alpha * acc (out_mode 0) or C += alpha * acc (1)
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const long n = col32 + 32 * j;
        if (n >= g.N) continue;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const long m = m0 + 32 * (wid % (NWV / 2)) + (r & 3) + 8 * (r >> 2) + 4 * lh;
            if (m >= g.M) continue;
            float* c = g.C + m * g.ldc + n;
            const float v = g.alpha * acc[j][r];
            *c = g.out_mode == 1 ? *c + v : v;
        }
    }
    if (col16 < g.N) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const long m = m0 + 16 * wid + 4 * lh16 + r;
            if (m >= g.M) continue;
            float* c = g.C + m * g.ldc + col16;
            const float v = g.alpha * acc16[r];
            *c = g.out_mode == 1 ? *c + v : v;
        }
    }
}

// g: TT problem (A = (K x M) row-contiguous, B = (N x K) k-contiguous), 128 < N <= 208; tiles_m = ceil(M / 64), split /
// k_chunk (multiple of 32) set by the caller; counters: adj_counter_ints(tiles_m, split) ints, zero between launches; slabs:
// adj_slab_floats(tiles_m, split) floats
inline long adj_groups(int split) { return split <= 16 ? 1 : (split + 7) / 8; }
inline long adj_counter_ints(long tiles_m, int split) { return tiles_m * (1 + adj_groups(split)); }
inline long adj_slab_floats(long tiles_m, int split, int bm = ADJ_BM) {
    return tiles_m * (split + (adj_groups(split) > 1 ? adj_groups(split) : 0)) * (long)(bm * ADJ_BN);
}

__global__ __launch_bounds__(256, 2) void gemm_adj_kernel(Args g) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    gemm_adj_body<0>(g, smem);
}
__global__ __launch_bounds__(256, 2) void gemm_adj_b16_kernel(Args g) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    gemm_adj_body<1>(g, smem);
}
__global__ __launch_bounds__(256, 2) void gemm_adj_f16x2_kernel(Args g) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    gemm_adj_body<2>(g, smem);
}

__global__ __launch_bounds__(512, 1) void gemm_adj128_f16x2_kernel(Args g) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    gemm_adj_body<2, 8>(g, smem);
}
__global__ __launch_bounds__(512, 1) void gemm_adj128_b16_kernel(Args g) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    gemm_adj_body<1, 8>(g, smem);
}
inline hipError_t launch_adj128_b16(const Args& g, hipStream_t s) {
    static NemoAttrOnce attr_once;
    if (attr_once.need()) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_adj128_b16_kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, adj_lds_bytes(8));
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(gemm_adj128_b16_kernel, dim3(g.tiles_m * g.split), dim3(512), adj_lds_bytes(8), s, g);
    return hipSuccess;
}
// the 128-row form of kind 2: tiles_m = ceil(M / 128), one workgroup of 512 threads per CU
inline hipError_t launch_adj128_f16x2(const Args& g, hipStream_t s) {
    static NemoAttrOnce attr_once;
    if (attr_once.need()) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_adj128_f16x2_kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, adj_lds_bytes(8));
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(gemm_adj128_f16x2_kernel, dim3(g.tiles_m * g.split), dim3(512), adj_lds_bytes(8), s, g);
    return hipSuccess;
}

// kind: 0 fp32 operands, 1 bf16 in memory, 2 fp16 piece planes (split % nseg == 0)
inline hipError_t launch_adj(const Args& g, hipStream_t s, int kind = 0) {
    static NemoAttrOnce attr_once[3];
    const void* fn = kind == 2 ? reinterpret_cast<const void*>(&gemm_adj_f16x2_kernel)
                   : kind == 1 ? reinterpret_cast<const void*>(&gemm_adj_b16_kernel) : reinterpret_cast<const void*>(&gemm_adj_kernel);
    if (attr_once[kind].need()) {
        hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, ADJ_LDS_BYTES);
        if (e != hipSuccess) return e;
    }
    if (kind == 2) hipLaunchKernelGGL(gemm_adj_f16x2_kernel, dim3(g.tiles_m * g.split), dim3(256), ADJ_LDS_BYTES, s, g);
    else if (kind == 1) hipLaunchKernelGGL(gemm_adj_b16_kernel, dim3(g.tiles_m * g.split), dim3(256), ADJ_LDS_BYTES, s, g);
    else hipLaunchKernelGGL(gemm_adj_kernel, dim3(g.tiles_m * g.split), dim3(256), ADJ_LDS_BYTES, s, g);
    return hipSuccess;
}

}  // namespace glds
