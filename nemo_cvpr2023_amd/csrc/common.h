// Shared helpers for the gfx950 kernels of libnemo_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define NEMO_OK 0
#define NEMO_EINVAL (-1)

#define NEMO_LAUNCH_CHECK()                         \
    do {                                            \
        hipError_t e_ = hipGetLastError();          \
        if (e_ != hipSuccess) return (int32_t)e_;   \
    } while (0)

static inline int nemo_cdiv(long a, long b) { return (int)((a + b - 1) / b); }

// 64-lane wavefront sum (all lanes receive the total).
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// Block-wide sum for blocks of up to 1024 threads; result valid in thread 0.
__device__ __forceinline__ float block_sum(float v, float* red /* >= 16 floats LDS */) {
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int nw = (blockDim.x + 63) >> 6;
    __syncthreads();
    if (lane == 0) red[wid] = v;
    __syncthreads();
    float t = 0.f;
    if (threadIdx.x == 0)
        for (int i = 0; i < nw; ++i) t += red[i];
    return t;
}
