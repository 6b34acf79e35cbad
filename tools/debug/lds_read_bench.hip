// LDS read micro-benchmark: cycles per wave-instruction of ds_read_b128 / ds_read_b64 / ds_read_b32 for the lane-address
// patterns the sparse skinning could use.  build: hipcc --offload-arch=gfx950 -O3 -o lds_read_bench lds_read_bench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));

template <int W>
__global__ __launch_bounds__(256) void k(const int* __restrict__ offs, float* out, unsigned long long* cyc, int iters) {
    extern __shared__ float lds[];
    for (int i = threadIdx.x; i < 16384; i += 256) lds[i] = (float)i;
    __syncthreads();
    const int off = offs[threadIdx.x & 63];          // byte offset of this lane
    const char* base = reinterpret_cast<const char*>(lds) + off;
    float acc = 0.f;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (W == 4) { f4 v = *reinterpret_cast<const f4*>(base + u * 4096); acc += v.x + v.y + v.z + v.w; }
            if (W == 2) { f2 v = *reinterpret_cast<const f2*>(base + u * 4096); acc += v.x + v.y; }
            if (W == 1) { acc += *reinterpret_cast<const float*>(base + u * 4096); }
        }
        asm volatile("" ::: "memory");
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * 256 + threadIdx.x] = acc;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

int main() {
    int* d_off; float* d_out; unsigned long long* d_cyc;
    hipMalloc(&d_off, 64 * 4); hipMalloc(&d_out, 256 * 256 * 4); hipMalloc(&d_cyc, 256 * 8);
    const int iters = 2000;
    const char* names[] = {"contiguous 16 B / lane", "segments: quarter g -> segment 7 g (256 B each), 16 B x (lane & 15)",
                           "row per sample: (lane & 15) x 1168 B + quarter x 48 B", "row per sample, stride 1160 B (290 dwords)",
                           "all lanes same address", "quarter segments, 4 B x (lane & 15)  (for the b32 form)",
                           "row per sample 1168 B, all quarters same joint"};
    for (int pat = 0; pat < 7; ++pat) {
        int h[64];
        for (int l = 0; l < 64; ++l) {
            const int l15 = l & 15, g = l >> 4;
            if (pat == 0) h[l] = 16 * l;
            if (pat == 1) h[l] = 7 * g * 256 + 16 * l15;
            if (pat == 2) h[l] = l15 * 1168 + g * 5 * 48;
            if (pat == 3) h[l] = l15 * 1160 + g * 5 * 48;
            if (pat == 4) h[l] = 0;
            if (pat == 5) h[l] = 7 * g * 64 + 4 * l15;
            if (pat == 6) h[l] = l15 * 1168;
        }
        hipMemcpy(d_off, h, sizeof(h), hipMemcpyHostToDevice);
        for (int W : {4, 2, 1}) {
            for (int blocks_per_cu : {1, 2}) {
                const int nb = 256 * blocks_per_cu;
                if (W == 4) hipLaunchKernelGGL(k<4>, dim3(nb), dim3(256), 65536, 0, d_off, d_out, d_cyc, iters);
                if (W == 2) hipLaunchKernelGGL(k<2>, dim3(nb), dim3(256), 65536, 0, d_off, d_out, d_cyc, iters);
                if (W == 1) hipLaunchKernelGGL(k<1>, dim3(nb), dim3(256), 65536, 0, d_off, d_out, d_cyc, iters);
                hipDeviceSynchronize();
                unsigned long long c[512];
                hipMemcpy(c, d_cyc, nb * 8, hipMemcpyDeviceToHost);
                double s = 0; for (int i = 0; i < nb; ++i) s += (double)c[i];
                // cycles per wave-instruction as seen by ONE wave; LDS cycles per instruction = that / (waves per CU)
                const double per = s / nb / (iters * 8.0);
                printf("pat %d W=%d B=%2d blocks/CU=%d: %.1f cycles per wave-instr (wave view) -> %.2f LDS cycles per instr\n", pat,
                       W, 4 * W, blocks_per_cu, per, per / (4.0 * blocks_per_cu));
            }
        }
        printf("   ^ %s\n", names[pat]);
    }
    return 0;
}
