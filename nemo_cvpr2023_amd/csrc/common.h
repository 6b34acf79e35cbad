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

// The > 64 KB dynamic-LDS opt-in (hipFuncSetAttribute) is per DEVICE: one flag per (call site, device), so that a process that drives
// several GPUs sets it on each (until round 6 a per-process flag set it on the first device only).
struct NemoAttrOnce {
    bool done[32] = {};
    bool need() {
        int d = 0;
        (void)hipGetDevice(&d);
        d &= 31;
        if (done[d]) return false;
        done[d] = true;
        return true;
    }
};

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

// ---- ordered (deterministic) accumulation across the blocks of a launch (round 5) ------------------------------------------
// Until round 4 per-view loss / camera / phase-network sums, bias column sums and the loss scalars were accumulated with
// float atomicAdd: the order of the additions, and with it the last bits of every sum, changed from run to run (1000 camera
// iterations at lr 0.1 amplified that to 4 % of the final camera loss).  Now every block deposits its partial value(s) in a
// region of a caller-owned scratch, takes a ticket, and the LAST-ARRIVING block sums the deposits in a fixed order and is the
// only writer of the outputs -- the scheme the fused mesh kernel and the split-K GEMMs already used.
//
// Host side: nemo_red_take(floats, tickets) hands out a region per launch -- bump allocation over the CALLER-OWNED arena the
// calling thread bound with nemo_reduce_ws_bind (include/nemo_hip.h, ABI 17; the bind / nemo_reduce_scratch_reset() at the top
// of every step rewinds it, so the launches of a step own distinct regions and a captured graph keeps the ones it was captured
// with).  Tickets are zero between launches (the last arriver resets its own).  When no arena is bound or it is exhausted the
// region is {nullptr, nullptr}, the kernels fall back to the atomics and nemo_reduce_fallbacks() counts it.
struct NemoRed { float* part; int* ticket; };
NemoRed nemo_red_take(size_t part_floats, int n_tickets);

#ifdef __HIPCC__
// A deposit: WRITE-THROUGH (sc1) store -- device-visible once the storing wave's vmcnt has drained, no L2 write-back fence (a
// release fence per block, `buffer_wbl2`, costs microseconds per block: the first version of this scheme took the 8 x 300 step
// from 1.38 to 1.54 ms and the minibatch-512 step from 0.56 to 0.99 ms).
__device__ __forceinline__ void nemo_red_put(float* p, float v) {
    __hip_atomic_store(reinterpret_cast<unsigned*>(p), __builtin_bit_cast(unsigned, v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void nemo_red_put(int* p, int v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// sixteen bytes at once (p 16-byte aligned): a 4-byte write-through store is one fabric write each, ~6x the time per byte
__device__ __forceinline__ void nemo_red_put4(float* p, float a, float b, float c, float d) {
    typedef float f4 __attribute__((ext_vector_type(4)));
    const f4 v = {a, b, c, d};
    asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
}

// Every thread of the block calls this after the block's deposits (nemo_red_put).  True in every thread of the LAST block to
// arrive on ticket `tk` (of `nblk`): the others' deposits are then visible to its plain loads.  Placement-independent
// hand-off (cdna_hip_programming.md G16, R1): write-through stores -> every wave drains vmcnt(0) -> barrier -> one-lane
// relaxed agent-scope ticket; consumer: one-lane agent acquire -> barrier -> plain loads.
__device__ __forceinline__ bool nemo_red_arrive(const NemoRed& r, int tk, int nblk, int* lds_flag) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0)
        *lds_flag = __hip_atomic_fetch_add(r.ticket + tk, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    if (*lds_flag != nblk - 1) return false;
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        __hip_atomic_store(r.ticket + tk, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);     // ready for the next launch
    }
    __syncthreads();
    return true;
}

// *out += scale-free sum over the blocks of a launch of the value `t` (valid in thread 0 of each block; block `blk` of `nblk`),
// added in block order by the last arriver (thread i takes deposits i, i + blockDim, ...; then block_sum's fixed tree).
// red: >= 16 floats of LDS, flag: one LDS int.  Every thread of the block must call it (barriers inside).
__device__ __forceinline__ void nemo_red_scalar(float t, float* out, const NemoRed& r, int blk, int nblk, float* red, int* flag) {
    if (!r.part) {
        if (threadIdx.x == 0 && t != 0.f) atomicAdd(out, t);
        return;
    }
    if (threadIdx.x == 0) nemo_red_put(r.part + blk, t);
    if (!nemo_red_arrive(r, 0, nblk, flag)) return;
    float a = 0.f;
    for (int i = threadIdx.x; i < nblk; i += blockDim.x) a += r.part[i];
    a = block_sum(a, red);
    if (threadIdx.x == 0) *out += a;
}

// Column sums: block (strip, chunk) holds the sums of 64 columns over its row chunk in threads 0 .. 63 (`t`; column n, valid
// while n < N); out[n] += the strip's chunks in chunk order.  Ticket `tk` is the strip's.
__device__ __forceinline__ void nemo_colsum_finish(float t, long n, long N, float* out, const NemoRed& rr, int strip, int n_strips, int chunk,
                                              int n_chunks, int tk, int* flag) {
    if (!rr.part) {
        if (threadIdx.x < 64 && n < N) atomicAdd(out + n, t);
        return;
    }
    float* dep = rr.part + ((size_t)strip * n_chunks) * 64;
    {
        __shared__ __attribute__((aligned(16))) float cdep[64];
        if (threadIdx.x < 64) cdep[threadIdx.x] = t;
        __syncthreads();
        if (threadIdx.x < 16)
            nemo_red_put4(dep + (size_t)chunk * 64 + 4 * threadIdx.x, cdep[4 * threadIdx.x], cdep[4 * threadIdx.x + 1],
                          cdep[4 * threadIdx.x + 2], cdep[4 * threadIdx.x + 3]);
    }
    if (!nemo_red_arrive(rr, tk, n_chunks, flag)) return;
    // all four waves of the last arriver take every fourth chunk (loads in flight), then wave 0 adds the four sums in order
    __shared__ float cfin[4][64];
    {
        const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
        float a = 0.f;
        int c = w;
        for (; c + 28 < n_chunks; c += 32) {             // eight deposits in flight (a round trip each otherwise), added in order
            float q[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) q[u] = dep[(size_t)(c + 4 * u) * 64 + l];
#pragma unroll
            for (int u = 0; u < 8; ++u) a += q[u];
        }
        for (; c < n_chunks; c += 4) a += dep[(size_t)c * 64 + l];
        cfin[w][l] = a;
    }
    __syncthreads();
    if (threadIdx.x < 64 && n < N) out[n] += (cfin[0][threadIdx.x] + cfin[1][threadIdx.x]) + (cfin[2][threadIdx.x] + cfin[3][threadIdx.x]);
}

// ---- scale records of the fp16 split-precision copies (include/nemo_hip.h nemo_gemm_xp fmt 2; csrc/gemm_xp.h) ------------------
namespace nemo_meta {
constexpr int META_FLOATS = 64, META_SLOTS = 32;            // a scale record: [0] scale, [2, 2 + META_SLOTS) absmax slots
// the absmax a record holds (every lane of the calling wave gets it)
__device__ __forceinline__ float meta_absmax(const float* m) {
    float v = m[2 + (threadIdx.x & (META_SLOTS - 1))];
#pragma unroll
    for (int off = META_SLOTS / 2; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, 64));
    return v;
}
__device__ __forceinline__ void meta_absmax_put(float* m, int slot, float mx) {
    float* p = m + 2 + (slot & (META_SLOTS - 1));
    // (the slot only grows: a value that is not above what is already there needs no atomic)
    if (mx > __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(reinterpret_cast<unsigned*>(p), __builtin_bit_cast(unsigned, mx));
}

}  // namespace nemo_meta

#endif
