// The absmax pass of the split-precision chain's scale records (include/nemo_hip.h nemo_absmax_multi), shared by its own kernel
// (gemm_xp.h) and by the first launch of an update step (pose.hip: the pass over the weights runs in further blocks of the phase
// kernel's launch instead of beside it on a second stream).
#pragma once
#include "common.h"
#include "../../include/nemo_hip.h"

namespace xp {

constexpr int MAX_CAST = NEMO_CAST_XP_MAX;
#ifndef ABSMAX_ROWS
#define ABSMAX_ROWS 16       // rows' loads in flight per thread: a 1000 x 1000 weight matrix is 31 rows per block in overwrite mode -- two round trips
#endif

// absmax slots of `meta` = max(themselves, max |src|) for a list of matrices (grid-stride over each; slots zero or previous maxima)
struct AbsmaxDesc { const float* src; long rows, cols, lds; float* meta; int block0; int overwrite; };
struct AbsmaxArgs { AbsmaxDesc d[MAX_CAST]; int n; };

// block `bid` of the `nblocks` absmax blocks of a launch (256 threads; a kernel of its own -- absmax_kernel, gemm_xp.h -- or the leading
// blocks of a step's first launch, pose.hip phase_embed_begin_kernel)
__device__ __forceinline__ void absmax_block(const AbsmaxArgs& a, const int bid, const int nblocks) {
    int di = 0;
#pragma unroll
    for (int i = 1; i < MAX_CAST; ++i)
        if (i < a.n && bid >= a.d[i].block0) di = i;
    const AbsmaxDesc& d = a.d[di];
    const int nb = (di + 1 < a.n ? a.d[di + 1].block0 : nblocks) - d.block0;
    // block b of the matrix's nb takes rows [b R / nb, (b + 1) R / nb): coalesced row segments, no index division per element
    // (the first form -- a flat grid-stride loop with e / cols, e % cols per element -- took 53 us for the chain's weights)
    const long b = (long)bid - d.block0;
    const long r0 = b * d.rows / nb, r1 = (b + 1) * d.rows / nb;
    float mx = 0.f;
    // ABSMAX_ROWS rows' loads in flight per thread (one load per round trip took 48 us for the chain's weights: 124 dependent trips; eight: 13.5)
    const float* src = d.src;
    const long lds = d.lds, cols = d.cols;
    // Every load is UNCONDITIONAL -- rows / columns beyond the range are clamped onto its last row / column (a maximum does not mind seeing
    // an element twice).  With a predicate per load (`r + u < r1 ? load : 0`) hipcc branches around each load and waits for it: one
    // round trip per row, 31 dependent trips for a 1000 x 1000 matrix in 32 blocks.
    if (r1 > r0 && cols > 0) {
        if ((lds & 3) == 0 && (cols & 3) == 0 && (reinterpret_cast<unsigned long long>(src) & 15) == 0) {
            const long c4n = cols >> 2;
            for (long c4 = threadIdx.x; c4 < c4n; c4 += 256)
                for (long r = r0; r < r1; r += ABSMAX_ROWS) {
                    float4 v[ABSMAX_ROWS];
#pragma unroll
                    for (int u = 0; u < ABSMAX_ROWS; ++u)
                        v[u] = *reinterpret_cast<const float4*>(src + min(r + u, r1 - 1) * lds + 4 * c4);
#pragma unroll
                    for (int u = 0; u < ABSMAX_ROWS; ++u)
                        mx = fmaxf(fmaxf(mx, fmaxf(fabsf(v[u].x), fabsf(v[u].y))), fmaxf(fabsf(v[u].z), fabsf(v[u].w)));
                }
        } else {
            // unaligned / narrow rows (nn.Linear(105, h), the head gradient, a strided view): thread = (row of four, column of 64); per pass
            // ABSMAX_ROWS / 2 row quads x one 64-column group, all loads in flight
            const long cg = (cols + 63) >> 6;
            const long tr = threadIdx.x >> 6, tc = threadIdx.x & 63;
            for (long g2 = 0; g2 < cg; ++g2) {
                const long c = min((g2 << 6) + tc, cols - 1);
                for (long r = r0; r < r1; r += 4 * (ABSMAX_ROWS / 2)) {
                    float v[ABSMAX_ROWS / 2];
#pragma unroll
                    for (int u = 0; u < ABSMAX_ROWS / 2; ++u) v[u] = src[min(r + 4 * u + tr, r1 - 1) * lds + c];
#pragma unroll
                    for (int u = 0; u < ABSMAX_ROWS / 2; ++u) mx = fmaxf(mx, fabsf(v[u]));
                }
            }
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off, 64));
    // (NaN: fmaxf drops it -- a NaN operand reaches the result through the pieces themselves)
    __shared__ float wmx[4];
    if ((threadIdx.x & 63) == 0) wmx[threadIdx.x >> 6] = mx;
    __syncthreads();
    if (threadIdx.x == 0) {
        const float bm = fmaxf(fmaxf(wmx[0], wmx[1]), fmaxf(wmx[2], wmx[3]));
        // overwrite: this matrix has exactly META_SLOTS blocks and block b OWNS slot b -- plain stores, nothing to zero beforehand
        if (d.overwrite) d.meta[2 + (bid - d.block0)] = bm;
        else nemo_meta::meta_absmax_put(d.meta, bid, bm);
    }
}


// descs -> kernel arguments; the number of blocks in `blocks`.  NEMO_EINVAL on a malformed list.
inline int absmax_fill(AbsmaxArgs& a, int n, const nemo_absmax_desc* descs, int& blocks) {
    if (n < 0 || n > MAX_CAST || (n && !descs)) return NEMO_EINVAL;
    a = AbsmaxArgs{};
    blocks = 0;
    int m = 0;
    for (int i = 0; i < n; ++i) {
        const nemo_absmax_desc& q = descs[i];
        if (q.rows < 0 || q.cols < 0 || !q.meta || q.lds < q.cols || ((q.rows && q.cols) && !q.src)) return NEMO_EINVAL;
        if ((q.rows == 0 || q.cols == 0) && !q.overwrite) continue;
        AbsmaxDesc& d = a.d[m++];
        d.src = q.src; d.rows = q.rows; d.cols = q.cols; d.lds = q.lds; d.meta = q.meta; d.block0 = blocks;
        d.overwrite = q.overwrite ? 1 : 0;
        long nb = (q.rows * q.cols + 4095) / 4096;          // ~16 elements per thread
        blocks += q.overwrite ? nemo_meta::META_SLOTS : (int)(nb < 1 ? 1 : (nb > 256 ? 256 : nb));
    }
    a.n = m;
    return NEMO_OK;
}

}  // namespace xp
