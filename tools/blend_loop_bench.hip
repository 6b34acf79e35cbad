// Micro-benchmark of the blend-shape k-step of the fused mesh kernel: 6 x v_mfma_f32_16x16x4_f32 per step
// with (optionally) one b96 buffer load into an 8-deep ring and two LDS operand reads, one wave per SIMD.
// Prints cycles per k-step (192 = MFMA-bound).   hipcc --offload-arch=gfx950 -O3 tools/blend_loop_bench.hip -o tools/blend_loop_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x3 __attribute__((ext_vector_type(3)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int LOADS, int LDSR, int SB>
__global__ __launch_bounds__(256, 2) void k(const float* P, int ldP, int iters, float* out, unsigned long long* ticks) {
    __shared__ float pf[2][16 * 226];
    const int tid = threadIdx.x, lane = tid & 63, l15 = lane & 15, g = lane >> 4;
    for (int i = tid; i < 2 * 16 * 226; i += 256) (&pf[0][0])[i] = 0.001f * (i & 255);
    __syncthreads();
    const __amdgpu_buffer_rsrc_t Prs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(P), 0, 224 * ldP * 4, 0x00020000);
    const int loff = (g * ldP + l15 * 3) * 4;
    const int kstride = 4 * ldP * 4;
    const float* pf0 = &pf[0][l15 * 226 + g];
    const float* pf1 = &pf[1][l15 * 226 + g];
    f32x4 vp[2][3];
    for (int b = 0; b < 2; ++b) for (int c = 0; c < 3; ++c) for (int r = 0; r < 4; ++r) vp[b][c][r] = 0.f;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);      // scalar: no waterfall loop around the buffer loads
    const int wave_tile = ((blockIdx.x * 4 + wid) & 31) * 192;
    const int wave_tile2 = ((blockIdx.x * 4 + wid) & 31) * 52 * 1024;
    u32x3 pa[8];
    // LOADS: 1 = row-major P (4 segments of 192 B per load), 2 = pre-tiled, 16 B per lane (1024 B contiguous),
    //        3 = pre-tiled, 12 B per lane (768 B contiguous)
    auto ld = [&](int ks) -> u32x3 {
        if (LOADS == 2) {
            u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(Prs, lane * 16, wave_tile2 + ks * 1024, 0);
            return u32x3{v[0], v[1], v[2]};
        } else if (LOADS == 4) {       // 4 B per lane only
            unsigned v = __builtin_amdgcn_raw_buffer_load_b32(Prs, lane * 4, wave_tile2 + ks * 1024, 0);
            return u32x3{v, v, v};
        } else if (LOADS == 5) {       // two k-steps per load: b128 every other step carries 2 x (a0,a1) .. (traffic test)
            if (ks & 1) return u32x3{0u, 0u, 0u};
            u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(Prs, lane * 16, wave_tile2 + ks * 1024, 0);
            return u32x3{v[0], v[1], v[2] ^ v[3]};
        } else if (LOADS == 3) {
            return __builtin_amdgcn_raw_buffer_load_b96(Prs, lane * 12, wave_tile2 + ks * 768, 0);
        }
        return __builtin_amdgcn_raw_buffer_load_b96(Prs, loff, wave_tile + ks * kstride, 0);
    };
    for (int u = 0; u < 8; ++u) pa[u] = ld(u);
    float pb[2][2];
    for (int u = 0; u < 2; ++u) { pb[u][0] = pf0[4 * u]; pb[u][1] = pf1[4 * u]; }
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
        for (int kk0 = 0; kk0 < 48; kk0 += 8) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const float a0 = __uint_as_float(pa[u][0]), a1 = __uint_as_float(pa[u][1]), a2 = __uint_as_float(pa[u][2]);
                const float b0 = pb[u & 1][0], b1 = pb[u & 1][1];
                if (LDSR) { pb[u & 1][0] = pf0[4 * ((kk0 + u + 2) % 48)]; pb[u & 1][1] = pf1[4 * ((kk0 + u + 2) % 48)]; }
                if (LOADS) pa[u] = ld((kk0 + 8 + u) % 48);
                vp[0][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, b0, vp[0][0], 0, 0, 0);
                vp[1][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, b1, vp[1][0], 0, 0, 0);
                vp[0][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, b0, vp[0][1], 0, 0, 0);
                vp[1][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, b1, vp[1][1], 0, 0, 0);
                vp[0][2] = __builtin_amdgcn_mfma_f32_16x16x4f32(a2, b0, vp[0][2], 0, 0, 0);
                vp[1][2] = __builtin_amdgcn_mfma_f32_16x16x4f32(a2, b1, vp[1][2], 0, 0, 0);
                if (SB) __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
    for (int b = 0; b < 2; ++b) for (int c = 0; c < 3; ++c) for (int r = 0; r < 4; ++r) s += vp[b][c][r];
    out[blockIdx.x * 256 + tid] = s;
    if (tid == 0 && blockIdx.x == 0) *ticks = t1 - t0;
}

template <int LOADS, int LDSR, int SB>
void run(const char* name, int blocks, const float* P, int ldP, int iters, float* out, unsigned long long* ticks) {
    hipLaunchKernelGGL((k<LOADS, LDSR, SB>), dim3(blocks), dim3(256), 0, 0, P, ldP, iters, out, ticks);
    hipDeviceSynchronize();
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipEventRecord(a);
    hipLaunchKernelGGL((k<LOADS, LDSR, SB>), dim3(blocks), dim3(256), 0, 0, P, ldP, iters, out, ticks);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    unsigned long long tk; hipMemcpy(&tk, ticks, 8, hipMemcpyDeviceToHost);
    printf("%-34s blocks=%4d  %8.1f us  cycles/k-step (block 0 wave 0) = %.1f\n", name, blocks, ms * 1e3, (double)tk / (48.0 * iters));
}

int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 200;
    const int ldP = 20688;
    float *P, *out; unsigned long long* ticks;
    hipMalloc(&P, (size_t)224 * ldP * 4); hipMemset(P, 0, (size_t)224 * ldP * 4);
    hipMalloc(&out, 1024 * 256 * 4); hipMalloc(&ticks, 8);
    for (int blocks : {256, 512}) {
        run<0, 0, 1>("mfma only", blocks, P, ldP, iters, out, ticks);
        run<0, 1, 1>("mfma + lds reads", blocks, P, ldP, iters, out, ticks);
        run<1, 0, 1>("mfma + ring loads", blocks, P, ldP, iters, out, ticks);
        run<1, 1, 1>("mfma + ring loads + lds reads", blocks, P, ldP, iters, out, ticks);
        run<1, 1, 0>("same, no sched_barrier", blocks, P, ldP, iters, out, ticks);
        run<2, 1, 1>("pre-tiled b128 + lds reads", blocks, P, ldP, iters, out, ticks);
        run<2, 1, 0>("pre-tiled b128, no sched_barrier", blocks, P, ldP, iters, out, ticks);
        run<3, 1, 1>("pre-tiled b96 + lds reads", blocks, P, ldP, iters, out, ticks);
        run<4, 1, 1>("pre-tiled b32 + lds reads", blocks, P, ldP, iters, out, ticks);
        run<5, 1, 1>("b128 every other k-step", blocks, P, ldP, iters, out, ticks);
    }
    return 0;
}
