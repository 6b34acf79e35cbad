// A CHAIN of dependent GEMMs as ONE persistent launch (round 4): the MotionNet forward  X -> H1 -> H2 -> H3 -> heads  and its
// backward (the dX chain with the parameter-gradient products as filler work) used to be ~14 dependent launches, each of
// which drained the machine before the next one filled it (~9.7 us fixed per launch + a 1.375-wave tail at 608 tiles on
// 512-768 block slots).  Here the workgroups of one launch pull (tile, K slice) items of ALL stages from one ticket
// counter, in an order in which every item only depends on items with smaller tickets, and wait on per-ROW-BAND arrival
// counters instead of kernel boundaries: a tile of layer l + 1 in row band r needs only the tiles_n(l) tiles of band r of
// layer l.  The tile itself is gemm_glds.h's (same LDS-DMA K loop, same epilogues, same split-K slabs / tickets).
//
// Replaces: the same operators as gemm_glds.h -- nn.Linear forward / backward of MotionNet
// (nemo/neural_motion_model.py:58-71, :130-148).
//
// Hand-off between workgroups of one launch (MI355X: per-XCD L2s are not coherent, a CU's vector L1 is never refreshed):
//   producer: fp32 results leave with write-through (sc1) stores (Args::wt) -> every wave s_waitcnt vmcnt(0) -> barrier ->
//             one lane: relaxed agent-scope fetch_add on the band's arrival counter;
//   consumer: one lane polls the counter (relaxed, s_sleep), barrier, then reads the band through sc1 LDS-DMA loads
//             (gemm_glds.h dma_piece<true>: past the vector L1) -- sc1 stores + sc1 loads need no fences.
// Deadlock freedom: tickets are handed out in list order by a device counter, an item waits only on items with smaller
// tickets, so the unfinished item with the smallest ticket is always held by a running workgroup that is not waiting.
// All counters (tickets, arrival counters) must be zero when the launch starts: the caller zero-fills them (in the
// step: with the accumulator arena, by the first launch of the step).
#pragma once
#include "gemm_glds.h"

namespace glds {

constexpr int CHAIN_MAX_STAGES = 8;
constexpr int CHAIN_MAX_BANDS = 192;            // row bands (64 rows) per stage: up to 12 288 rows
// int counters in the scratch, ALL ZERO when the launch starts (the caller's job: the step's first launch zero-fills them
// with its accumulator arena; nothing here resets them): [16 x] ticket counter of XCD list x (one 64-byte line each),
// [128 + s] tiles of stage s done, [160 + s * CHAIN_MAX_BANDS + band] tiles of (stage s, row band) done
constexpr int CHAIN_CTL_INTS = 160 + CHAIN_MAX_STAGES * CHAIN_MAX_BANDS;

struct ChainStage {
    Args g;             // (counters / slabs: this stage's own region of the scratch when split > 1)
    int per_band;       // items per row band: tiles_n * max(split, 1)
    int layout;         // 0: A (M x K) k-contiguous, B (N x K) k-contiguous (forward); 1: A k-contiguous, B (K x N) (dX);
                        // 2: A (K x M), B (K x N) (parameter gradients)
    int dep;            // stage this one reads its A operand from (-1: kernel inputs only)
    int dep_mode;       // 1: row band tm of `dep` must be complete; 2: every tile of `dep`
    int signal;         // 1: count finished tiles in (some later stage waits on this one)
};

// Work lists: ONE PER XCD.  Row band tm of every stage belongs to list tm % 8, and a workgroup pulls from the list of the
// XCD it runs on (s_getreg XCC_ID; placement is a speed matter only): the 16 column tiles of a band then share the band's A
// rows in ONE L2, a band of layer l + 1 is read on the XCD that wrote it, and eight ticket words are pulled by ~96
// workgroups each instead of one word by 768 (one word serves ~88 dequeues per microsecond).  A list holds its bands stage
// by stage, band-major; a workgroup whose own list is exhausted takes from the other lists (loads first: a fetch_add only
// where something is left).
struct ChainArgs {
    ChainStage st[CHAIN_MAX_STAGES];
    int n_stages;
    int list_len[8];
    int* ctl;           // CHAIN_CTL_INTS ints, zero at launch
};

template <int LAYOUT, bool ASC1>
__device__ __forceinline__ bool chain_tile(const Args& g, const Coord co, float* smem) {
    // A operands are read past the vector L1 (sc1) in every stage: stage outputs of this launch, or inputs (0 - 3 % slower)
    if constexpr (LAYOUT == 0) return gemm_glds_tile<64, 64, 32, 32, 32, true, true, 3, true, 0, ASC1>(g, co, smem);
    else if constexpr (LAYOUT == 1) return gemm_glds_tile<64, 64, 32, 32, 32, true, false, 3, true, 0, ASC1>(g, co, smem);
    else return gemm_glds_tile<64, 64, 32, 32, 32, false, false, 3, true, 0, ASC1>(g, co, smem);
}

constexpr int CHAIN_LDS_BYTES = 3 * (64 + 64) * BK * 4 + 16;

template <int LAYOUTS, bool ASC1 = true>      // bit l set: the chain has stages of layout l
__global__ __launch_bounds__(256, 2) void gemm_chain_kernel(ChainArgs ca) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    // (all LDS in ONE array: a second __shared__ object makes hipcc drain the LDS-DMA pipeline in front of every ds_read)
    volatile int* s_ctl = reinterpret_cast<volatile int*>(smem + 3 * (64 + 64) * BK);
    int* const ctl = ca.ctl;
    unsigned xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    const int home = (int)(xcc & 7u);
    for (;;) {
        __syncthreads();                     // (the previous item's LDS stages / its s_ctl are no longer read)
        if (threadIdx.x == 0) {
            int list = home;
            int t = __hip_atomic_fetch_add(ctl + 16 * home, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (t >= ca.list_len[home]) {
                t = -1;
                for (int d = 1; d < 8 && t < 0; ++d) {
                    const int y = (home + d) & 7;
                    if (__hip_atomic_load(ctl + 16 * y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < ca.list_len[y]) {
                        const int q = __hip_atomic_fetch_add(ctl + 16 * y, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        if (q < ca.list_len[y]) { t = q; list = y; }
                    }
                }
            }
            s_ctl[0] = t;
            s_ctl[1] = list;
        }
        __syncthreads();
        int t = __builtin_amdgcn_readfirstlane(s_ctl[0]);
        const int list = __builtin_amdgcn_readfirstlane(s_ctl[1]);
        if (t < 0) break;
        // ticket -> (stage, band, column tile, K slice) of list `list`
        int s = 0;
#pragma unroll
        for (int q = 0; q < CHAIN_MAX_STAGES - 1; ++q) {
            if (q == s && q + 1 < ca.n_stages) {
                const int tmq = ca.st[q].g.tiles_m;
                const int items = (tmq > list ? (tmq - list + 7) / 8 : 0) * ca.st[q].per_band;
                if (t >= items) { t -= items; s = q + 1; }
            }
        }
        const ChainStage& S = ca.st[s];
        const int split = S.g.split > 1 ? S.g.split : 1;
        const int tm = list + 8 * (t / S.per_band), r = t % S.per_band;
        const int tn = r / split, slice = r % split;
        if (S.dep >= 0) {
            if (threadIdx.x == 0) {
                const int* p = S.dep_mode == 1 ? ctl + 160 + S.dep * CHAIN_MAX_BANDS + tm : ctl + 128 + S.dep;
                const int need = S.dep_mode == 1 ? ca.st[S.dep].g.tiles_n : ca.st[S.dep].g.n_tiles;
                while (__hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < need) __builtin_amdgcn_s_sleep(8);
            }
            __syncthreads();
        }
        const Coord co{tm * S.g.tiles_n + tn, tm, tn, slice, split};
        bool wrote;
        if ((LAYOUTS & 1) && (S.layout == 0 || LAYOUTS == 1)) wrote = chain_tile<0, ASC1>(S.g, co, smem);
        else if ((LAYOUTS & 2) && (S.layout == 1 || !(LAYOUTS & 4))) wrote = chain_tile<1, ASC1>(S.g, co, smem);
        else wrote = chain_tile<2, ASC1>(S.g, co, smem);
        if (S.signal && wrote) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // every storing wave: its sc1 stores have left
            __syncthreads();
            if (threadIdx.x == 0) {
                __hip_atomic_fetch_add(ctl + 160 + s * CHAIN_MAX_BANDS + tm, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (S.signal & 2) __hip_atomic_fetch_add(ctl + 128 + s, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
}

template <int LAYOUTS, bool ASC1 = true>
hipError_t launch_chain(const ChainArgs& ca, int blocks, hipStream_t s) {
    static bool attr_set = false;
    auto kern = &gemm_chain_kernel<LAYOUTS, ASC1>;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                           CHAIN_LDS_BYTES);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), CHAIN_LDS_BYTES, s, ca);
    return hipSuccess;
}

}  // namespace glds
