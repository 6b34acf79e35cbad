// fp32 MFMA GEMM for SKINNY problems -- a few hundred rows on one side (the one-instance shard of the sharded fit:
// 300 samples against 1000 x 1000 layers), where a 64 x 64 tile grid cannot fill 256 CUs and the in-launch split-K of
// gemm_glds.h pays a cross-block hand-off (write-through slabs, device-scope ticket, slab read-back: ~4.5 us of a
// ~19 us launch, tools/gemm_skinny_dev).  Here the K split is INSIDE the block:
//   * output tile (32 NI) x (32 NJ); the W waves of a block each own a K slice of the whole tile and multiply straight
//     from global memory (no LDS staging: nothing is shared between waves, a wave's operand fragment is exactly what
//     one buffer load returns -- see the k permutation below), D steps of 8 k in flight per wave;
//   * (optionally K is ALSO cut across blocks -- Args::split slices, write-through slabs and a ticket per tile as in
//     gemm_glds.h -- for very long K over few tiles: the blend-shape adjoint dPF = dVP P^T of a one-instance shard);
//   * the W partial tiles meet in LDS (W NI NJ 4 KiB), every wave sums 16 NI NJ / W accumulator rows in wave order
//     (deterministic) and runs the epilogue.
// Same operations as gemm_glds.h / gemm.hip: nn.Linear forward / backward of MotionNet and VPoser
// (nemo/neural_motion_model.py:58-71,130-148; human_body_prior/models/vposer_model.py:69-88).
//
// Fragments.  v_mfma_f32_32x32x2_f32 takes, per lane, one A element (row = lane & 31, k = lane >> 5) and one B
// element (col = lane & 31, same k).  A step covers 8 consecutive k with 4 MFMAs; MFMA j takes k = k0 + 4 h + j from
// the lanes of half h = lane >> 5 for BOTH operands (a permutation of the step's k, the same on both sides):
//   image K operand (k contiguous in memory): the lane's four elements are ONE buffer_load_dwordx4;
//   image M operand (rows contiguous):        four buffer_load_dword, each 2 x 128 B fully coalesced.
// Bounds come from the buffer descriptors (extent = last valid byte, glds::extents): rows past M / N and -- image M --
// k rows past K read as zero; pipeline slots past a wave's K slice are pointed out of range and multiply zeros.
// Only the last partial step of an image-K operand (k >= K inside a row) is masked by hand.
#pragma once
#include "gemm_glds.h"

namespace skinny {

using glds::Args;
using glds::f32x16;

// VEC: an image-K operand whose rows are 16-byte aligned (base and ld): one dwordx4 per step; otherwise four dwords
// (nn.Linear weights with in_features = 105, views that start 3 floats into a row).
template <bool KC, bool VEC>
struct Src {
    __amdgpu_buffer_rsrc_t rs;
    int voff;
    unsigned ld4;           // bytes per k row (image M) / per element step (image K: 4)
    __device__ __forceinline__ void init(const float* base, unsigned bytes, long ld, long row0, int lane) {
        rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base), 0, (int)bytes, 0x00020000);
        const int r = lane & 31, h = lane >> 5;
        voff = KC ? (int)(((row0 + r) * ld + 4 * h) * 4) : (int)((4 * h * ld + row0 + r) * 4);
        ld4 = KC ? 4u : (unsigned)(ld * 4);
    }
    // the lane's four operands of the step starting at k0; live == false: all zeros (out-of-range address)
    __device__ __forceinline__ void load(long k0, bool live, float (&f)[4]) const {
        const int vo = live ? voff : (int)0xfffffff0u;
        const unsigned so = live ? (unsigned)k0 * ld4 : 0u;
        if constexpr (KC && VEC) {
            const glds::f32x4 v = __builtin_bit_cast(glds::f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, vo, (int)so, 0));
            f[0] = v[0]; f[1] = v[1]; f[2] = v[2]; f[3] = v[3];
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j)
                f[j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, vo, (int)(live ? so + j * ld4 : 0u), 0));
        }
    }
};

// grid: 8 * tiles_m * ceil(tiles_n / 8) blocks; block b -> XCD b & 7 takes the column tiles tn = xcd (mod 8) and all
// row tiles of them, so that an XCD's L2 holds 1/8 of the B operand (the weights) and the skinny A operand.
// Block tile (32 NI) x (32 NJ): NI x NJ accumulators per wave, NI + NJ fragments per step.
template <bool AKC, bool BKC, bool AV, bool BV, int NI, int NJ, int W, int D>
__global__ __launch_bounds__(64 * W) void gemm_skinny_kernel(Args g) {
    extern __shared__ __attribute__((aligned(16))) float red[];       // [W][NI * NJ][16][64]
    constexpr int NA = NI * NJ;
    // g.split > 1: K is also cut across blocks (long K over few tiles: the blend-shape adjoint of a one-instance shard);
    // slice-major block ids, a tile's slices 8 * tiles_m * ceil(tiles_n / 8) blocks apart: the same XCD.
    const int per_slice = (int)gridDim.x / g.split;
    const int slice = (int)blockIdx.x / per_slice, bid = (int)blockIdx.x - slice * per_slice, xcd = bid & 7, bi = bid >> 3;
    // fewer than 8 column tiles: a compact grid (the XCD-striped one would launch up to 7/8 empty blocks, each of which
    // still waits for a CU with this kernel's LDS allocation free -- with a 112 KiB reduction buffer that serialises them
    // behind the working blocks: 10x the run time of a 32 x 224 tile)
    const bool compact = g.tiles_n < 8;
    const int tm = compact ? bid / g.tiles_n : bi % g.tiles_m, tn = compact ? bid % g.tiles_n : (bi / g.tiles_m) * 8 + xcd;
    if (tn >= g.tiles_n) return;
    const long m0 = (long)tm * (32 * NI), n0 = (long)tn * (32 * NJ);
    const int lane = threadIdx.x & 63;
    const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);

    Src<AKC, AV> sa[NI];
    Src<BKC, BV> sb[NJ];
#pragma unroll
    for (int i = 0; i < NI; ++i) sa[i].init(g.A, g.a_bytes, g.lda, m0 + 32 * i, lane);
#pragma unroll
    for (int j = 0; j < NJ; ++j) sb[j].init(g.B, g.b_bytes, g.ldb, n0 + 32 * j, lane);

    // K slice of this wave, in steps of 8 k; a partial last step is kept out of the pipeline
    // (g.k_chunk, the K range of a slice, is a multiple of 8)
    const long kbeg = (long)slice * g.k_chunk, kend = min(g.K, kbeg + g.k_chunk);
    const int s0 = (int)(kbeg / 8), nfull = (int)(kend / 8) - s0;
    const int s_beg = s0 + (int)((long)nfull * wid / W), s_end = s0 + (int)((long)nfull * (wid + 1) / W);
    const int rounds = (s_end - s_beg + D - 1) / D;

    f32x16 acc[NI][NJ];
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    float fa[D][NI][4], fb[D][NJ][4];
    auto request = [&](int d, int s) {
#pragma unroll
        for (int i = 0; i < NI; ++i) sa[i].load(8L * s, s < s_end, fa[d][i]);
#pragma unroll
        for (int j = 0; j < NJ; ++j) sb[j].load(8L * s, s < s_end, fb[d][j]);
        __builtin_amdgcn_sched_barrier(0);          // slot order = issue order, whichever way the loop is entered (the
    };                                              // compiler counts vmcnt per slot instead of draining the ring)
#pragma unroll
    for (int d = 0; d < D; ++d) request(d, s_beg + d);
    for (int r = 0; r < rounds; ++r) {
#pragma unroll
        for (int d = 0; d < D; ++d) {
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int i = 0; i < NI; ++i)
#pragma unroll
                    for (int j = 0; j < NJ; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[d][i][q], fb[d][j][q], acc[i][j], 0, 0, 0);
            // (the scheduler would otherwise gather a round's loads behind its last MFMA: nothing in flight under them)
            __builtin_amdgcn_sched_barrier(0);
            request(d, s_beg + (r + 1) * D + d);
        }
    }
    if (wid == W - 1 && (g.K & 7) && kend == g.K) {
        // the partial last step: k = k0 + 4 h + q >= K contributes nothing
        const long k0 = 8L * (s0 + nfull);
        const int h = lane >> 5;
        float ta[NI][4], tb[NJ][4];
#pragma unroll
        for (int i = 0; i < NI; ++i) sa[i].load(k0, true, ta[i]);
#pragma unroll
        for (int j = 0; j < NJ; ++j) sb[j].load(k0, true, tb[j]);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const bool in = k0 + 4 * h + q < g.K;
#pragma unroll
            for (int i = 0; i < NI; ++i)
#pragma unroll
                for (int j = 0; j < NJ; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(in ? ta[i][q] : 0.f, in ? tb[j][q] : 0.f, acc[i][j], 0, 0, 0);
        }
    }

    // ---- the W partial tiles meet in LDS; wave w finishes accumulator rows [w RPW, (w + 1) RPW) of the NA x 16
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) red[((wid * NA + i * NJ + j) * 16 + r) * 64 + lane] = acc[i][j][r];
    __syncthreads();
    constexpr int RPW = 16 * NA / W;                // accumulator rows per wave
    static_assert(16 * NA % W == 0, "accumulator rows must divide over the waves");
    const int lr = lane & 31, lh = lane >> 5;
    float part[RPW];
#pragma unroll
    for (int u = 0; u < RPW; ++u) {
        const int ar = wid * RPW + u;
        float v = 0.f;
#pragma unroll
        for (int w = 0; w < W; ++w) v += red[((w * NA + ar / 16) * 16 + ar % 16) * 64 + lane];   // wave order: deterministic
        part[u] = v;
    }
    if (g.split > 1) {
        // publish the slice's tile (write-through), take a ticket; the last arriver sums all slices in slice order
        const int tile = tn * g.tiles_m + tm;
        float* slab = g.slabs + ((size_t)tile * g.split + slice) * (size_t)(NA * 1024);
#pragma unroll
        for (int u = 0; u < RPW; ++u) {
            float* dst = slab + (wid * RPW + u) * 64 + lane;
            asm volatile("global_store_dword %0, %1, off sc1" ::"v"(dst), "v"(part[u]) : "memory");
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        int* flag = reinterpret_cast<int*>(red);
        if (threadIdx.x == 0)
            *flag = __hip_atomic_fetch_add(g.counters + tile, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
        if (*flag != g.split - 1) return;
        if (threadIdx.x == 0) {
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            __hip_atomic_store(g.counters + tile, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        __syncthreads();
        const float* base = g.slabs + (size_t)tile * g.split * (size_t)(NA * 1024);
        // slice-major: the RPW loads of a slice are independent and in flight together (row-major, a wide tile's 28 rows
        // x `split` slices were one dependent load chain per row: ~100 us for 25 slices of a 32 x 224 tile)
#pragma unroll
        for (int u = 0; u < RPW; ++u) part[u] = 0.f;
        for (int sl = 0; sl < g.split; ++sl) {            // fixed order: deterministic
            const float* sb_ = base + (size_t)sl * (NA * 1024) + wid * RPW * 64 + lane;
#pragma unroll
            for (int u = 0; u < RPW; ++u) part[u] += sb_[u * 64];
        }
    }
#pragma unroll
    for (int u = 0; u < RPW; ++u) {
        const int ar = wid * RPW + u, a = ar / 16, r = ar % 16, i = a / NJ, j = a % NJ;
        float v = part[u];
        const long n = n0 + 32 * j + lr;
        const long m = m0 + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (n >= g.N || m >= g.M) continue;
        v = g.alpha * v + (g.bias ? g.bias[n] : 0.f);
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

template <bool AKC, bool BKC, bool AV, bool BV, int NI, int NJ, int W, int D>
hipError_t launch(Args g, hipStream_t s) {
    constexpr int lds = W * NI * NJ * 16 * 64 * (int)sizeof(float);
    auto kern = &gemm_skinny_kernel<AKC, BKC, AV, BV, NI, NJ, W, D>;
    if (lds > 64 * 1024) {
        static NemoAttrOnce attr_once;
        if (attr_once.need()) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
            if (e != hipSuccess) return e;
        }
    }
    g.tiles_m = (int)((g.M + 32 * NI - 1) / (32 * NI));
    g.tiles_n = (int)((g.N + 32 * NJ - 1) / (32 * NJ));
    if (g.split < 1) g.split = 1;
    const int blocks = (g.tiles_n < 8 ? g.tiles_m * g.tiles_n : 8 * g.tiles_m * ((g.tiles_n + 7) / 8)) * g.split;
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(64 * W), lds, s, g);
    return hipSuccess;
}

}  // namespace skinny
