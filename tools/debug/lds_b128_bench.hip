// ds_read_b128 cycles per wave-instruction for the address pattern  row(l15) * STRIDE + g * GOFF  (bytes), 4 waves / block,
// 1 or 2 blocks per CU.   hipcc --offload-arch=gfx950 -O3 tools/debug/lds_b128_bench.hip -o /tmp/lds_b128 && /tmp/lds_b128
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void k(int stride, int goff, int iters, unsigned long long* out, unsigned* sink) {
    extern __shared__ unsigned char lds[];
    for (int i = threadIdx.x; i < 16384; i += 256) reinterpret_cast<unsigned*>(lds)[i] = i;
    __syncthreads();
    const int lane = threadIdx.x & 63, l15 = lane & 15, g = lane >> 4;
    const unsigned char* p = lds + l15 * stride + g * goff;
    u32x4 acc = {0, 0, 0, 0};
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const u32x4 v = *reinterpret_cast<const u32x4*>(p + u * 64);
            acc += v;
        }
        asm volatile("" ::: "memory");
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
    if (acc.x + acc.y + acc.z + acc.w == 0x12345) sink[0] = 1;
}
int main() {
    unsigned long long* out; unsigned* sink;
    hipMalloc(&out, 8 * 2048); hipMalloc(&sink, 4);
    const int iters = 200;
    hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    int cfgs[][2] = {{784, 16}, {768, 16}, {800, 16}, {464, 16}, {272, 16}, {16, 256}, {64, 16}, {784 + 32, 16}, {784, 0}, {1040, 16}, {528, 16}, {784, 64}, {832, 16}, {1296, 16}};
    for (auto& c : cfgs)
        for (int blocks : {256, 512}) {
            hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 65536, 0, c[0], c[1], iters, out, sink);
            hipDeviceSynchronize();
            std::vector<unsigned long long> h(blocks);
            hipMemcpy(h.data(), out, 8 * blocks, hipMemcpyDeviceToHost);
            double s = 0; for (auto v : h) s += v;
            printf("stride %5d goff %4d blocks %3d: %.1f cycles per ds_read_b128 per wave (x %d waves/CU -> %.1f LDS cycles each)\n", c[0], c[1], blocks,
                   s / blocks / (iters * 16.0), blocks / 64, s / blocks / (iters * 16.0) / (blocks / 64));
        }
    return 0;
}
