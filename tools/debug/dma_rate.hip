// What a CU can pull from L2 / MALL / HBM into LDS, by mechanism (round 5: the large-tile bf16 GEMM's K loop runs at ~40 % of
// the matrix pipe and moves ~40 GB/s per CU -- is that the LDS-DMA path's ceiling?).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/debug/dma_rate.hip -o tools/debug/dma_rate
// One workgroup per CU (140 KiB of LDS), W waves; every wave streams 1 KiB pieces (16 rows x 64 B at row stride `ld`, the GEMM's
// stage image) of a per-workgroup source region of R bytes, again and again, D pieces in flight per wave.
//   mode 0: LDS-DMA (buffer_load_dwordx4 ... lds)      mode 1: global_load_dwordx4 -> VGPR -> ds_write_b128
//   mode 2: global_load_dwordx4 -> VGPR only (no LDS)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)
typedef int i32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void dma_piece(i32x4 rs, unsigned lds_byte, int voff, unsigned soff) {
    unsigned keep;
    soff = __builtin_amdgcn_readfirstlane(soff);
    lds_byte = __builtin_amdgcn_readfirstlane(lds_byte);
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %2, %3, %4 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "s"(lds_byte), "v"(voff), "s"(rs), "s"(soff) : "memory");
}

template <int MODE, int D, int SEG>
__global__ __launch_bounds__(768) void stream_kernel(const char* src, long region, long ld, int pieces_per_wave, unsigned* sink) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63, wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), nw = blockDim.x >> 6;
    const char* base = src + (long)blockIdx.x * region;
    const unsigned long long b = reinterpret_cast<unsigned long long>(base);
    const i32x4 rs = {(int)(unsigned)b, (int)(unsigned)((b >> 32) & 0xffffu), (int)region, 0x00020000};
    constexpr int LPS = SEG / 16, RPP = 1024 / SEG;          // lanes per row segment, rows per piece
    const int voff = (int)((lane / LPS) * ld + (lane % LPS) * 16);
    const __amdgpu_buffer_rsrc_t rsb = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(base), 0, (int)region, 0x00020000);
    const unsigned lds0 = (unsigned)reinterpret_cast<unsigned long long>(smem) + wid * (D * 1024);
    // piece i of this wave: rows 16 (i * nw + wid) ..., wrapped inside the region
    const unsigned rows = (unsigned)(region / ld);           // (a power of two)
    const unsigned ld_u = (unsigned)ld, segs = ld_u / SEG;   // (powers of two)
    const unsigned rsh = 31 - __builtin_clz(rows);
    i32x4 v[D];
    unsigned acc = 0;
    for (int i = 0; i < pieces_per_wave; i += D) {
#pragma unroll
        for (int u = 0; u < D; ++u) {
            const unsigned piece = (unsigned)(i + u) * nw + wid;
            const unsigned pr = piece * RPP;
            const unsigned so = (pr & (rows - 1)) * ld_u + ((pr >> rsh) & (segs - 1)) * SEG;
            if (MODE == 0) dma_piece(rs, lds0 + u * 1024, voff, so);
            else v[u] = __builtin_bit_cast(i32x4, __builtin_amdgcn_raw_buffer_load_b128(rsb, voff, (int)so, 0));
        }
        if (MODE == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else {
#pragma unroll
            for (int u = 0; u < D; ++u) {
                if (MODE == 1) *reinterpret_cast<i32x4*>(smem + wid * (D * 1024) + u * 1024 + lane * 16) = v[u];
                else acc += v[u][0] ^ v[u][1] ^ v[u][2] ^ v[u][3];
            }
        }
    }
    if (MODE == 1) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); acc = smem[threadIdx.x]; }
    if (acc == 0x12345678u) sink[0] = acc;
}

template <int MODE, int D, int SEG>
static float run(int blocks, int waves, const char* src, long region, long ld, int ppw, unsigned* sink, hipEvent_t e0, hipEvent_t e1) {
    auto k = &stream_kernel<MODE, D, SEG>;
    static bool set = false;
    if (!set) { CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, 140 * 1024)); set = true; }
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL(k, dim3(blocks), dim3(64 * waves), 140 * 1024, 0, src, region, ld, ppw, sink);
    CK(hipEventRecord(e0));
    for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(k, dim3(blocks), dim3(64 * waves), 140 * 1024, 0, src, region, ld, ppw, sink);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    return ms / 5;
}

int main() {
    const long total = 1L << 30;
    char* src; CK(hipMalloc(&src, total)); CK(hipMemset(src, 1, total));
    unsigned* sink; CK(hipMalloc(&sink, 64));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const long ld = 2048;
    printf("per-CU streaming rate into LDS (m0: LDS-DMA) / VGPRs (m2), 1 KiB pieces of 1024 / seg rows x seg bytes, row stride %ld B, 8 in flight per wave; GB/s per CU (chip TB/s)\n", ld);
    for (int blocks : {24, 256})
        for (long region : {256L << 10, 4L << 20})
            for (int waves : {4, 12}) {
                if (blocks * region > total) continue;
                const int ppw = 4096;
                const double bytes = (double)blocks * waves * ppw * 1024.0;
                printf("blocks %3d region %5ld KiB waves %2d:", blocks, region >> 10, waves);
#define R(MODE, D, SEG) { const float ms = run<MODE, D, SEG>(blocks, waves, src, region, ld, ppw, sink, e0, e1); printf("  m%d seg%d %6.1f (%.2f)", MODE, SEG, bytes / ms / 1e6 / blocks, bytes / ms / 1e9); }
                R(0, 8, 64) R(0, 8, 128) R(0, 8, 256) R(0, 8, 1024) R(2, 8, 64) R(2, 8, 128) R(2, 8, 1024)
                printf("\n");
            }
    return 0;
}
