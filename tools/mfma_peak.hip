// Measures the sustained issue rate of the fp32 MFMAs (v_mfma_f32_32x32x2_f32, v_mfma_f32_16x16x4_f32)
// with NACC independent accumulators per wave and W waves per SIMD, and the shader clock under that load
// (s_memtime ticks vs wall clock).  Build: hipcc --offload-arch=gfx950 -O3 tools/mfma_peak.hip -o /tmp/mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NACC>
__global__ __launch_bounds__(256) void k32(float* out, int iters, unsigned long long* ticks) {
    f32x16 acc[NACC];
    for (int a = 0; a < NACC; ++a) for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
    float x = threadIdx.x * 1e-3f, y = blockIdx.x * 1e-3f;
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 16 / NACC; ++u)
#pragma unroll
            for (int a = 0; a < NACC; ++a) acc[a] = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, acc[a], 0, 0, 0);
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
    for (int a = 0; a < NACC; ++a) for (int r = 0; r < 16; ++r) s += acc[a][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) *ticks = t1 - t0;
}

template <int NACC>
__global__ __launch_bounds__(256) void k16(float* out, int iters, unsigned long long* ticks) {
    f32x4 acc[NACC];
    for (int a = 0; a < NACC; ++a) for (int r = 0; r < 4; ++r) acc[a][r] = 0.f;
    float x = threadIdx.x * 1e-3f, y = blockIdx.x * 1e-3f;
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 16 / NACC; ++u)
#pragma unroll
            for (int a = 0; a < NACC; ++a) acc[a] = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, acc[a], 0, 0, 0);
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
    for (int a = 0; a < NACC; ++a) for (int r = 0; r < 4; ++r) s += acc[a][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) *ticks = t1 - t0;
}

template <typename F>
void run(const char* name, F launch, int blocks, int iters, double flop_per_mfma, float* out, unsigned long long* ticks) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    launch(blocks, iters);                        // warm
    hipDeviceSynchronize();
    hipEventRecord(a);
    launch(blocks, iters);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    unsigned long long tk; hipMemcpy(&tk, ticks, 8, hipMemcpyDeviceToHost);
    const double mfmas_per_wave = 16.0 * iters;
    const double tf = flop_per_mfma * mfmas_per_wave * 4.0 * blocks / (ms * 1e-3) / 1e12;
    printf("%-28s blocks=%4d  %8.1f us  %7.1f TFLOP/s   counter ticks/MFMA(wave 0)=%.1f  tick rate=%.0f MHz\n", name, blocks,
           ms * 1e3, tf, (double)tk / mfmas_per_wave, tk / (ms * 1e3));
}

int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 2000;
    float* out; unsigned long long* ticks;
    hipMalloc(&out, 4096 * 256 * 4); hipMalloc(&ticks, 8);
    for (int rep = 0; rep < 2; ++rep)
        for (int blocks : {256, 512, 1024}) {
            run("32x32x2 1 acc (dependent)", [&](int b, int it) { hipLaunchKernelGGL(k32<1>, dim3(b), dim3(256), 0, 0, out, it, ticks); }, blocks, iters, 4096.0, out, ticks);
            run("32x32x2 4 acc", [&](int b, int it) { hipLaunchKernelGGL(k32<4>, dim3(b), dim3(256), 0, 0, out, it, ticks); }, blocks, iters, 4096.0, out, ticks);
            run("16x16x4 1 acc (dependent)", [&](int b, int it) { hipLaunchKernelGGL(k16<1>, dim3(b), dim3(256), 0, 0, out, it, ticks); }, blocks, iters, 2048.0, out, ticks);
            run("16x16x4 4 acc", [&](int b, int it) { hipLaunchKernelGGL(k16<4>, dim3(b), dim3(256), 0, 0, out, it, ticks); }, blocks, iters, 2048.0, out, ticks);
        }
    // short bursts, like one GEMM of the step (tens of microseconds)
    for (int it : {20, 50, 200})
        run("32x32x2 4 acc, short", [&](int b, int i2) { hipLaunchKernelGGL(k32<4>, dim3(b), dim3(256), 0, 0, out, i2, ticks); }, 256, it, 4096.0, out, ticks);
    return 0;
}
