// What a device-wide barrier costs inside ONE launch on MI355X (the alternative to a kernel boundary in a replayed graph):
// 256 / 512 co-resident blocks, ticket + generation barrier with device-scope atomics, BOUNDED spins (a failed barrier ends
// the kernel and is reported instead of hanging the box).
//   hipcc --offload-arch=gfx950 -O3 tools/debug/grid_barrier_bench.hip -o tools/debug/grid_barrier_bench
#include <hip/hip_runtime.h>
#include <cstdio>
__device__ __forceinline__ bool grid_barrier(unsigned* count, unsigned* gen, unsigned nblocks, unsigned& my_gen) {
    __syncthreads();
    bool ok = true;
    if (threadIdx.x == 0) {
        const unsigned target = my_gen + 1;
        const unsigned prev = __hip_atomic_fetch_add(count, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        if (prev == nblocks - 1) {
            __hip_atomic_store(count, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(gen, target, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            long spins = 0;
            while (__hip_atomic_load(gen, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) != target) {
                __builtin_amdgcn_s_sleep(1);
                if (++spins > 20000000) { ok = false; break; }
            }
        }
    }
    my_gen += 1;
    __syncthreads();
    return ok;
}
__global__ __launch_bounds__(256) void k(unsigned* count, unsigned* gen, int rounds, float* data, int work, int* fail) {
    unsigned my_gen = 0;
    float acc = 0.f;
    for (int r = 0; r < rounds; ++r) {
        for (int i = 0; i < work; ++i) acc += data[(blockIdx.x * 256 + threadIdx.x + i * 65536) & 0xfffff];   // a little memory work
        if (!grid_barrier(count, gen, gridDim.x, my_gen)) { if (threadIdx.x == 0) atomicAdd(fail, 1); return; }
    }
    if (acc == 12345.f) data[0] = acc;
}
int main() {
    unsigned *count, *gen; float* data; int* fail;
    hipMalloc(&count, 4); hipMalloc(&gen, 4); hipMalloc(&data, 4 << 20); hipMalloc(&fail, 4);
    hipMemset(data, 0, 4 << 20);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int blocks : {256, 512})
        for (int work : {0, 4}) {
            float ms[2];
            for (int pass = 0; pass < 2; ++pass) {
                const int rounds = pass ? 1100 : 100;
                hipMemset(count, 0, 4); hipMemset(gen, 0, 4); hipMemset(fail, 0, 4);
                hipEventRecord(e0);
                hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, count, gen, rounds, data, work, fail);
                hipEventRecord(e1); hipEventSynchronize(e1);
                hipEventElapsedTime(&ms[pass], e0, e1);
            }
            int f = 0; hipMemcpy(&f, fail, 4, hipMemcpyDeviceToHost);
            printf("blocks %d work %d: %.2f us per (work + barrier) round, failed blocks %d\n", blocks, work, (ms[1] - ms[0]) * 1e3 / 1000.0, f);
        }
    return 0;
}
