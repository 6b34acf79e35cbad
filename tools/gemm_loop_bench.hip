// Micro-benchmark of the K-tile iteration of the 64x64 fp32 GEMM tile (4 waves, one 32x32 accumulator each):
// 16 x v_mfma_f32_32x32x2_f32 per wave and iteration plus, switchable, the LDS operand reads (8 x ds_read_b128),
// the barrier, the LDS stores (4 x ds_write_b128) and the global loads (4 x dwordx4) of the real kernel.
// Prints cycles per iteration (1024 = MFMA-bound).  hipcc --offload-arch=gfx950 -O3 -fno-slp-vectorize tools/gemm_loop_bench.hip -o tools/gemm_loop_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int READS, int BARRIER, int STORES, int LOADS, int NACC, int AHEAD, int MAP>
__global__ __launch_bounds__(256, 2) void k(const float* A, long lda, int iters, float* out, unsigned long long* ticks) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, l31 = lane & 31, lhi = lane >> 5;
    const int wm = wid >> 1, wn = wid & 1;
    float* As = smem; float* Bs = smem + 2 * 64 * 36;
    for (int i = tid; i < 4 * 64 * 36; i += 256) smem[i] = 0.001f * (i & 127);
    __syncthreads();
    f32x16 acc[2];
    for (int a = 0; a < 2; ++a) for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
    // staging: thread -> (row, k chunk) of the 64 x 32 tiles, as in the kernel
    // MAP 0: a wave stores 8 rows x 128 B (two rows per 16 lanes: their b128 bank ranges overlap by 4 banks);
    // MAP 1: 16 rows x 64 B (16 lanes = 16 rows of one chunk: 36 r mod 64 are 16 distinct multiples of 4)
    const int kk = MAP ? 4 * ((tid >> 4) & 3) : 4 * (tid % 8), r0 = MAP ? (tid & 15) + 16 * (tid >> 6) : tid / 8;
    const int d2 = MAP ? 16 : 32 * 36;               // second float4 of an operand: chunk + 4 (same row) or row + 32
    const float* pa0 = A + (long)((blockIdx.x * 64 + r0) % 4096) * lda + kk;
    const float* pa1 = MAP ? pa0 + 16 : pa0 + 32 * lda;
    const float* pb0 = A + (long)((blockIdx.x * 64 + 2048 + r0) % 4096) * lda + kk;
    const float* pb1 = MAP ? pb0 + 16 : pb0 + 32 * lda;
    float4 v[4] = {};
    int cur = 0;
    float4 fa[2], fb[2];
    auto fetch = [&](int buf, int q, int slot) {
        fa[slot] = *reinterpret_cast<const float4*>(As + buf * 64 * 36 + (wm * 32 + l31) * 36 + 8 * q + 4 * lhi);
        fb[slot] = *reinterpret_cast<const float4*>(Bs + buf * 64 * 36 + (wn * 32 + l31) * 36 + 8 * q + 4 * lhi);
    };
    auto mf = [&](int slot) {
        const float a4[4] = {fa[slot].x, fa[slot].y, fa[slot].z, fa[slot].w};
        const float b4[4] = {fb[slot].x, fb[slot].y, fb[slot].z, fb[slot].w};
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int a = NACC == 2 ? (s & 1) : 0;
            acc[a] = __builtin_amdgcn_mfma_f32_32x32x2f32(READS ? a4[s] : 1.f + s, READS ? b4[s] : 2.f, acc[a], 0, 0, 0);
        }
    };
    if (READS) fetch(0, 0, 0);
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
        if (STORES) {
            float* d = As + (cur ^ 1) * 64 * 36 + r0 * 36 + kk;
            *reinterpret_cast<float4*>(d) = v[0]; *reinterpret_cast<float4*>(d + d2) = v[1];
            d = Bs + (cur ^ 1) * 64 * 36 + r0 * 36 + kk;
            *reinterpret_cast<float4*>(d) = v[2]; *reinterpret_cast<float4*>(d + d2) = v[3];
        }
        if (LOADS) {
            const long off = (long)((it * 32) & 1023);
            v[0] = *reinterpret_cast<const float4*>(pa0 + off); v[1] = *reinterpret_cast<const float4*>(pa1 + off);
            v[2] = *reinterpret_cast<const float4*>(pb0 + off); v[3] = *reinterpret_cast<const float4*>(pb1 + off);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (AHEAD) {
            if (READS) fetch(cur, 1, 1); __builtin_amdgcn_sched_barrier(0);
            mf(0); __builtin_amdgcn_sched_barrier(0);
            if (READS) fetch(cur, 2, 0); __builtin_amdgcn_sched_barrier(0);
            mf(1); __builtin_amdgcn_sched_barrier(0);
            if (READS) fetch(cur, 3, 1); __builtin_amdgcn_sched_barrier(0);
            mf(0); __builtin_amdgcn_sched_barrier(0);
            if (BARRIER) __syncthreads();
            if (READS) fetch(cur ^ 1, 0, 0); __builtin_amdgcn_sched_barrier(0);
            mf(1); __builtin_amdgcn_sched_barrier(0);
        } else {
            if (READS) fetch(cur, 0, 0);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                if (READS && q + 1 < 4) fetch(cur, q + 1, (q + 1) & 1);
                mf(q & 1);
            }
            if (BARRIER) __syncthreads();
        }
        cur ^= 1;
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    float s = v[0].x + v[1].y + v[2].z + v[3].w;
    for (int a = 0; a < 2; ++a) for (int r = 0; r < 16; ++r) s += acc[a][r];
    out[blockIdx.x * 256 + tid] = s;
    if (tid == 0 && blockIdx.x == 0) *ticks = t1 - t0;
}

template <int READS, int BARRIER, int STORES, int LOADS, int NACC, int AHEAD, int MAP = 0>
void run(const char* name, int blocks, const float* A, int iters, float* out, unsigned long long* ticks) {
    const int lds = 4 * 64 * 36 * 4;
    auto launch = [&]() { hipLaunchKernelGGL((k<READS, BARRIER, STORES, LOADS, NACC, AHEAD, MAP>), dim3(blocks), dim3(256), lds, 0, A, 1024L, iters, out, ticks); };
    launch(); hipDeviceSynchronize();
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipEventRecord(a); launch(); hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    unsigned long long tk; hipMemcpy(&tk, ticks, 8, hipMemcpyDeviceToHost);
    printf("%-52s blocks=%4d  %8.1f us  cycles/iteration (block 0) = %7.1f   wall/iteration = %.3f us\n", name, blocks, ms * 1e3, (double)tk / iters, ms * 1e3 / iters);
}

int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 400;
    float *A, *out; unsigned long long* ticks;
    hipMalloc(&A, 4096L * 1024 * 4); hipMemset(A, 0, 4096L * 1024 * 4);
    hipMalloc(&out, 2048 * 256 * 4); hipMalloc(&ticks, 8);
    for (int blocks : {256, 512, 1024}) {
        run<0, 0, 0, 0, 1, 0>("mfma only, 1 accumulator", blocks, A, iters, out, ticks);
        run<0, 0, 0, 0, 2, 0>("mfma only, 2 accumulators", blocks, A, iters, out, ticks);
        run<1, 0, 0, 0, 1, 0>("+ lds reads (as the compiler schedules them)", blocks, A, iters, out, ticks);
        run<1, 0, 0, 0, 1, 1>("+ lds reads one group ahead (pinned)", blocks, A, iters, out, ticks);
        run<1, 1, 0, 0, 1, 0>("+ reads + barrier", blocks, A, iters, out, ticks);
        run<1, 1, 1, 0, 1, 0>("+ reads + barrier + lds stores", blocks, A, iters, out, ticks);
        run<1, 1, 1, 1, 1, 0>("+ reads + barrier + stores + global loads (all)", blocks, A, iters, out, ticks);
        run<1, 1, 1, 0, 1, 0, 1>("+ reads + barrier + lds stores, 16-row store map", blocks, A, iters, out, ticks);
        run<1, 1, 1, 1, 1, 0, 1>("all, 16-row store map", blocks, A, iters, out, ticks);
        run<1, 1, 1, 1, 2, 0>("all, 2 accumulators", blocks, A, iters, out, ticks);
        run<1, 1, 1, 1, 1, 1>("all, reads ahead + mid barrier", blocks, A, iters, out, ticks);
        run<1, 1, 1, 1, 2, 1>("all, reads ahead + mid barrier, 2 accumulators", blocks, A, iters, out, ticks);
    }
    return 0;
}
