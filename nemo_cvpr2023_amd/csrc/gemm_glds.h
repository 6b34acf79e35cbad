// fp32 MFMA GEMM core whose operand tiles travel global -> LDS by LDS-DMA (buffer_load_dwordx4 ... lds): no staging
// VGPRs, no ds_write pass, K tiles in flight across the per-tile barrier under a counted vmcnt.  Included by
// gemm.hip (product) and tools/gemm_glds_dev.hip (stand-alone numerics + timing harness).
//
// Replaces, like gemm.hip's first-generation kernel: nn.Linear forward / backward of MotionNet and VPoser
// (nemo/neural_motion_model.py:58-71,130-148; human_body_prior/models/vposer_model.py:69-88) and the blend-shape
// adjoint dPF = dVP P^T (human_body_prior/body_model/lbs.py:229-233 backward).  Exact fp32 arithmetic
// (v_mfma_f32_32x32x2_f32 / v_mfma_f32_16x16x4_f32 are fmaf chains).
//
// LDS images of a (ROWS x 32) operand tile -- an LDS-DMA piece is 64 lanes x 16 B landing at base + 16*lane, so the
// image is lane-linear per piece and any swizzle has to be put on the SOURCE address:
//   image K (source k-contiguous, rows of 128 B): [row][8 chunks of 16 B], chunk c of row r stored at slot
//           c ^ ((r >> 1) & 7).  A piece = 8 rows.  An MFMA operand group (4 consecutive k of one row) is ONE
//           ds_read_b128; the 16 lanes a b128 read services together hold 16 rows whose (parity, (r>>1)&7) are all
//           different -> 16 distinct 16-byte bank quads, conflict-free.
//   image M (source row-contiguous, i.e. [k][ROWS]): linear.  A piece = 256/ROWS k-rows.  An operand group is four
//           ds_read_b32 (32 consecutive rows of one k: 32 consecutive banks).
// k permutation (as in gemm.hip): MFMA step j of group q takes k = KG*q + 4*h + j from the lanes of k-slice h
// (32x32x2: h = lane>>5, KG = 8; 16x16x4: h = lane>>4, KG = 16) for BOTH operands, so a lane's four operands of a
// group are contiguous in k.
//
// Bounds: both operands are read through buffer descriptors whose extent is the operand's last valid byte, so
// rows beyond M / N and -- for image M -- k rows beyond K return zeros without predicates.  Only a partial last K
// tile of an image-K operand (k >= K inside a row is the next row's data, not out of bounds) goes through a masked
// register-staged path, once per K slice that contains the end of K.
#pragma once
#include <type_traits>
#include "common.h"

namespace glds {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// One LDS-DMA piece: 64 lanes x 16 B from buffer `rs` at byte offset voff (per lane) + soff (uniform) to LDS bytes
// [lds_byte, lds_byte + 1024).  Issued from inline asm on purpose: hipcc orders every later ds_read behind an LDS-DMA
// it can see with s_waitcnt vmcnt(0) (it cannot prove that the stage being read and the stage being filled differ),
// which would drain the tiles in flight at the first operand fetch of every iteration.  The kernel orders the DMA
// against its readers itself (counted vmcnt + barrier).  M0 carries the LDS address; it is compiler-reserved, so it
// is saved and restored inside the statement.
__device__ __forceinline__ void dma_piece(i32x4 rs, unsigned lds_byte, int voff, unsigned soff) {
    unsigned keep;
    soff = __builtin_amdgcn_readfirstlane(soff);          // uniform by construction; keeps the operands in SGPRs
    lds_byte = __builtin_amdgcn_readfirstlane(lds_byte);
    asm volatile(
        "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %2, %3, %4 offen lds\n\ts_mov_b32 m0, %0"
        : "=&s"(keep)
        : "s"(lds_byte), "v"(voff), "s"(rs), "s"(soff)
        : "memory");
}

struct Args {
    const float* A; const float* B; float* C;
    const float* bias; const float* mask;
    float* slabs; int* counters;
    long M, N, K, lda, ldb, ldc, ldmask;
    long k_chunk;           // K range per slice (multiple of 32)
    int tiles_m, tiles_n, n_tiles, split;
    int t0;                 // tiles [0, t0) whole, tiles [t0, n_tiles) in `split` K slices
    int act, mask_mode, out_mode;
    float alpha;
    unsigned a_bytes, b_bytes;      // buffer extents (last valid byte + 1) of the two operands
    int xcd_order;                  // 1: whole-tile launch of 8 * ceil(tiles_m / 8) * tiles_n blocks in XCD-aware order
    // optional bf16 copies of the RESULT (after bias / activation / mask), written by the epilogue: Cb[m][n] (row stride
    // ldcb) and its transpose CbT[n][m] (row stride ldcbt) -- what the next GEMMs of a bf16-in-memory chain read as
    // k-contiguous operands (round 3, BF16 == 2)
    unsigned short* Cb = nullptr;
    unsigned short* CbT = nullptr;
    long ldcb = 0, ldcbt = 0;
    // BF16 == 2 only: the activation mask read from a bf16 copy (sign and zero survive the rounding) instead of `mask`;
    // C may be NULL when only the bf16 copies of the result are wanted (hidden activations of a bf16-in-memory chain)
    const unsigned short* mask16 = nullptr;
    long ldmask16 = 0;
    // (32 x 32 accumulators; fp32 operands since round 5 too) column sums of the RESULT (after bias / activation / mask, out_mode 0)
    // per 32-row wave band -- row
    // (tile_m * (BM / 32) + band) of colsum[.][ldcs] gets the band's sum of every column: the bias gradient of a layer is
    // then a sum over 2 ceil(M / 64) short rows instead of a pass over the M x N result, which no longer has to exist in
    // fp32 at all.  One writer per element, no atomics: deterministic.
    float* colsum = nullptr;
    long ldcs = 0;
    // gemm_adj.h, split-precision form (round 5): the K slices are dealt over `nseg` (operand plane of A, operand plane of B)
    // pairs -- slices [q split / nseg, (q + 1) split / nseg) read A + seg_a[q] and B + seg_b[q] (offsets in floats) over the whole K
    long seg_a[3] = {0, 0, 0}, seg_b[3] = {0, 0, 0};
    int nseg = 1;
};

constexpr int BK = 32;

template <int N> __device__ __forceinline__ void wait_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// ------------------------------------------------------------------------------------------ staging
// One operand: ROWS x 32 tile, image K (KC) or image M.
template <int ROWS, bool KC, int NW = 4>      // NW: waves of the workgroup that issue the pieces (4; 8 in the 128-row adjoint tile)
struct Operand {
    static constexpr int FLOATS = ROWS * BK;
    static constexpr int NI = ROWS / 8;                 // 1-KiB DMA pieces per tile
    static constexpr int CPR = ROWS / 4;                // image M: 16-byte chunks per k row
    static_assert(KC || 64 % CPR == 0, "image M needs ROWS in {16, 32, 64, 128, 256}");
    i32x4 rs;               // raw buffer descriptor: base, stride 0, extent in bytes, dword format
    int voff;               // per-lane byte offset of this wave's pieces (same for all of them, see below)
    long ld;

    // A wave issues pieces u = wid, wid + 4, ...: (u & 1) == (wid & 1), so the image-K swizzle term that depends
    // on the piece index is a per-wave constant and ONE lane offset serves all of a wave's pieces.
    __device__ __forceinline__ void init(const float* base, unsigned bytes, long ld_, int lane, int wid) {
        const unsigned long long b = reinterpret_cast<unsigned long long>(base);
        rs = i32x4{(int)(unsigned)b, (int)(unsigned)((b >> 32) & 0xffffu), (int)bytes, 0x00020000};
        ld = ld_;
        if (KC) {
            const int row = lane >> 3, chunk = (lane & 7) ^ (lane >> 4) ^ (4 * (wid & 1));
            voff = (int)((row * ld + 4 * chunk) * 4);
        } else {
            const int k = lane / CPR, c = lane % CPR;
            voff = (int)((k * ld + 4 * c) * 4);
        }
    }

    // pieces of the tile whose first row is row0 and first k is k0, into the LDS bytes starting at `tile_byte`
    __device__ __forceinline__ void dma(unsigned tile_byte, long row0, long k0, int wid) const {
#pragma unroll
        for (int u0 = 0; u0 < NI; u0 += NW) {
            const int u = u0 + wid;
            if (NI % NW != 0 && u >= NI) break;            // (wave-uniform)
            const long so = KC ? ((row0 + 8 * u) * ld + k0) * 4 : ((k0 + (64 / CPR) * u) * ld + row0) * 4;
            dma_piece(rs, tile_byte + u * 1024, voff, (unsigned)so);
        }
    }
    // pieces a wave issues per tile (waves with wid < NI % 4 issue one more when NI % 4 != 0)
    static constexpr int PER_WAVE = NI / NW;
    // the i-th of this wave's pieces alone (interleaved issue: one piece between two MFMAs)
    __device__ __forceinline__ void dma_one(unsigned tile_byte, long row0, long k0, int wid, int i) const {
        const int u = NW * i + wid;
        const long so = KC ? ((row0 + 8 * u) * ld + k0) * 4 : ((k0 + (64 / CPR) * u) * ld + row0) * 4;
        dma_piece(rs, tile_byte + u * 1024, voff, (unsigned)so);
    }

    // Masked register-staged fill of a PARTIAL last K tile (image K only; image M gets zeros from the descriptor).
    __device__ __forceinline__ void fill_tail(float* tile, const float* base, long row0, long k0, long rows,
                                              long K) const {
        static_assert(KC, "tail fill is for image K");
        for (int idx = threadIdx.x; idx < ROWS * 8; idx += 64 * NW) {
            const int r = idx >> 3, c = idx & 7;
            const long gr = row0 + r, gk = k0 + 4 * c;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (gr < rows) {
                const float* p = base + gr * ld + gk;
                if (gk + 3 < K) v = *reinterpret_cast<const float4*>(p);
                else {
                    if (gk < K) v.x = p[0];
                    if (gk + 1 < K) v.y = p[1];
                    if (gk + 2 < K) v.z = p[2];
                }
            }
            *reinterpret_cast<float4*>(tile + r * BK + ((c ^ ((r >> 1) & 7)) << 2)) = v;
        }
    }

    // bf16 path: the eight consecutive k (16 s + 8 h .. + 7) of `row`, rounded to bf16 (RNE) -- one operand of
    // v_mfma_f32_32x32x16_bf16 (k-step s of the tile, lane half h)
    static __device__ __forceinline__ bf16x8 fetch8(const float* tile, int row, int s, int h) {
        float f[8];
        if (KC) {
            const int c0 = 4 * s + 2 * h, sw = (row >> 1) & 7;
            const float4 a = *reinterpret_cast<const float4*>(tile + row * BK + ((c0 ^ sw) << 2));
            const float4 b = *reinterpret_cast<const float4*>(tile + row * BK + (((c0 + 1) ^ sw) << 2));
            f[0] = a.x; f[1] = a.y; f[2] = a.z; f[3] = a.w; f[4] = b.x; f[5] = b.y; f[6] = b.z; f[7] = b.w;
        } else {
#pragma unroll
            for (int i = 0; i < 8; ++i) f[i] = tile[(16 * s + 8 * h + i) * ROWS + row];
        }
        bf16x8 v;
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = (__bf16)f[i];
        return v;
    }

    // bf16-in-memory tiles (BF16 == 2): a row of the tile is 64 bf16 = 128 B = the same eight 16-byte chunks; chunk
    // 2 s + h IS the operand of k-step s (16 k) for lane half h -- one ds_read_b128, no conversion
    static __device__ __forceinline__ bf16x8 fetch16(const float* tile, int row, int s, int h) {
        static_assert(KC, "bf16-in-memory operands are k-contiguous");
        const int c = 2 * s + h;
        return *reinterpret_cast<const bf16x8*>(tile + row * BK + ((c ^ ((row >> 1) & 7)) << 2));
    }

    // four operands (MFMA steps 0..3) of group q for `row`, k-slice h
    template <int KG>      // 8 (32x32x2) or 16 (16x16x4)
    static __device__ __forceinline__ void fetch(const float* tile, int row, int q, int h, float (&f)[4]) {
        if (KC) {
            const int c = (KG / 4) * q + h;
            const float4 v = *reinterpret_cast<const float4*>(tile + row * BK + ((c ^ ((row >> 1) & 7)) << 2));
            f[0] = v.x; f[1] = v.y; f[2] = v.z; f[3] = v.w;
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) f[j] = tile[(KG * q + 4 * h + j) * ROWS + row];
        }
    }
};

// ------------------------------------------------------------------------------------------ kernel
// 256 threads = 4 waves as (BM/WM) x (BN/WN); a wave owns WM x WN of the block tile as (WM/T) x (WN/T) accumulators of
// the T x T MFMA (T = 32: v_mfma_f32_32x32x2_f32, T = 16: v_mfma_f32_16x16x4_f32).  NST LDS stages.
// BF16: operands are rounded to bf16 on their way from LDS into the matrix cores (v_mfma_f32_32x32x16_bf16, fp32
// accumulate; memory stays fp32 on both sides) -- BASELINE configs[2].
template <int BM, int BN, int WM, int WN, int T, bool AKC, bool BKC, int NST, bool SPREAD = false, int BF16 = 0>
__device__ __forceinline__ void gemm_glds_body(const Args& g, const int bid, float* smem) {
    constexpr int TM = WM / T, TN = WN / T, WGN = BN / WN;
    static_assert((BM / WM) * (BN / WN) == 4, "4 waves");
    constexpr int KG = T == 32 ? 8 : 16, NQ = BK / KG, AR = T * T / 64;      // AR: accumulator registers
    using OA = Operand<BM, AKC>;
    using OB = Operand<BN, BKC>;
    constexpr int STAGE = OA::FLOATS + OB::FLOATS;
    constexpr bool UNIFORM = (OA::NI % 4 == 0) && (OB::NI % 4 == 0);
    static_assert(UNIFORM || NST == 2, "counted vmcnt needs the same number of pieces on every wave");
    // SPREAD: the pieces of the tile being requested are issued one at a time BETWEEN the MFMAs of the tile being
    // multiplied (an LDS-DMA issue holds the wave's instruction stream for ~100 cycles, more than one MFMA covers: a
    // burst of 4-8 of them at the top of an iteration leaves the matrix pipe idle for most of it).  The last piece
    // then leaves late in iteration t, so the tile needs two more iterations to land: NST >= 3.
    static_assert(!SPREAD || (UNIFORM && NST >= 3 && T == 32), "spread issue: uniform piece counts, >= 3 stages");
    static_assert(!BF16 || T == 32, "bf16 path uses the 32x32x16 MFMA");
    constexpr int G = OA::PER_WAVE + OB::PER_WAVE;                           // pieces per wave and tile (UNIFORM)
    typedef float accv __attribute__((ext_vector_type(AR)));
    const bool whole = bid < g.t0 || g.xcd_order;
    const int n_tail = g.n_tiles - g.t0;
    int tile = whole ? bid : g.t0 + (bid - g.t0) % n_tail;
    const int slice = whole ? 0 : (bid - g.t0) / n_tail;
    const int split = whole ? 1 : g.split;
    int tm = tile % g.tiles_m, tn = tile / g.tiles_m;
    if (g.xcd_order) {
        // Large launches (thousands of tiles, no K split): block b runs on XCD b % 8 (observed dispatch order; only
        // speed depends on it).  Give every XCD its own row blocks of the A operand -- tm = 8 i + xcd -- and walk all
        // column tiles of a row block before the next one, so the ~96 blocks an XCD holds at a time share 6 row blocks of
        // A and the whole (small) B in that XCD's L2 instead of every L2 streaming all of A once per column tile.
        const int xcd = bid & 7, i = bid >> 3;
        tm = (i / g.tiles_n) * 8 + xcd;
        tn = i % g.tiles_n;
        if (tm >= g.tiles_m) return;
        tile = tn * g.tiles_m + tm;
    }
    const long m0 = (long)tm * BM, n0 = (long)tn * BN;
    const long kbeg = whole ? 0 : (long)slice * g.k_chunk;
    const long kend = whole ? g.K : min(g.K, kbeg + g.k_chunk);
    const long klen = kend > kbeg ? kend - kbeg : 0;
    const int nt = (int)((klen + BK - 1) / BK);
    // a partial last tile needs the masked path only for image-K operands
    const bool tail = (AKC || BKC) && (klen % BK) != 0;
    const int nfull = tail ? nt - 1 : nt;

    const int lane = threadIdx.x & 63;
    const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = wid / WGN, wn = wid % WGN;
    const int lr = T == 32 ? (lane & 31) : (lane & 15), lh = T == 32 ? (lane >> 5) : (lane >> 4);

    accv acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < AR; ++r) acc[i][j][r] = 0.f;

    const unsigned smem_byte = (unsigned)reinterpret_cast<unsigned long long>(smem);     // LDS offset of the stages
    OA oa; OB ob;
    oa.init(g.A, g.a_bytes, g.lda, lane, wid);
    ob.init(g.B, g.b_bytes, g.ldb, lane, wid);
    auto issue = [&](int t) {
        const unsigned st = smem_byte + (unsigned)((t % NST) * STAGE * 4);
        oa.dma(st, m0, kbeg + (long)t * BK, wid);
        ob.dma(st + OA::FLOATS * 4, n0, kbeg + (long)t * BK, wid);
    };
    // t_next >= 0: tile to request while multiplying (SPREAD)
    auto compute = [&](const float* st, int t_next) {
        const float* as = st;
        const float* bs = st + OA::FLOATS;
        const unsigned nb = smem_byte + (unsigned)(((t_next < 0 ? 0 : t_next) % NST) * STAGE * 4);
        const long nk = kbeg + (long)(t_next < 0 ? 0 : t_next) * BK;
        if constexpr (BF16 == 2) {
            // operands already bf16 in memory: a K tile is 64 k = four MFMA k-steps of 16; the tile being requested goes
            // out in four quarters behind them
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                bf16x8 fa[TM], fb[TN];
#pragma unroll
                for (int i = 0; i < TM; ++i) fa[i] = OA::fetch16(as, wm * WM + i * 32 + lr, ks, lh);
#pragma unroll
                for (int j = 0; j < TN; ++j) fb[j] = OB::fetch16(bs, wn * WN + j * 32 + lr, ks, lh);
                if constexpr (SPREAD) {
                    if (t_next >= 0) {
#pragma unroll
                        for (int p = (ks * G) / 4; p < ((ks + 1) * G) / 4; ++p) {
                            if (p < OA::PER_WAVE) oa.dma_one(nb, m0, nk, wid, p);
                            else ob.dma_one(nb + OA::FLOATS * 4, n0, nk, wid, p - OA::PER_WAVE);
                        }
                    }
                }
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);
            }
        } else if constexpr (BF16) {
            // two MFMA k-steps of 16 per tile; the tile being requested goes out in two halves behind them
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                bf16x8 fa[TM], fb[TN];
#pragma unroll
                for (int i = 0; i < TM; ++i) fa[i] = OA::fetch8(as, wm * WM + i * 32 + lr, ks, lh);
#pragma unroll
                for (int j = 0; j < TN; ++j) fb[j] = OB::fetch8(bs, wn * WN + j * 32 + lr, ks, lh);
                if constexpr (SPREAD) {
                    if (t_next >= 0) {
#pragma unroll
                        for (int p = ks * (G / 2); p < (ks == 0 ? G / 2 : G); ++p) {
                            if (p < OA::PER_WAVE) oa.dma_one(nb, m0, nk, wid, p);
                            else ob.dma_one(nb + OA::FLOATS * 4, n0, nk, wid, p - OA::PER_WAVE);
                        }
                    }
                }
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);
            }
        } else if constexpr (T == 32) {
            float fa[2][TM][4], fb[2][TN][4];
#pragma unroll
            for (int i = 0; i < TM; ++i) OA::template fetch<KG>(as, wm * WM + i * 32 + lr, 0, lh, fa[0][i]);
#pragma unroll
            for (int j = 0; j < TN; ++j) OB::template fetch<KG>(bs, wn * WN + j * 32 + lr, 0, lh, fb[0][j]);
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                if (q + 1 < NQ) {
#pragma unroll
                    for (int i = 0; i < TM; ++i)
                        OA::template fetch<KG>(as, wm * WM + i * 32 + lr, q + 1, lh, fa[(q + 1) & 1][i]);
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        OB::template fetch<KG>(bs, wn * WN + j * 32 + lr, q + 1, lh, fb[(q + 1) & 1][j]);
                }
                __builtin_amdgcn_sched_barrier(0);     // the next group's operand reads stay AHEAD of this group's MFMAs
#pragma unroll
                for (int s = 0; s < 4; ++s) {
#pragma unroll
                    for (int i = 0; i < TM; ++i)
#pragma unroll
                        for (int j = 0; j < TN; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[q & 1][i][s], fb[q & 1][j][s],
                                                                             acc[i][j], 0, 0, 0);
                    if constexpr (SPREAD) {
                        // piece p of this wave's G goes out after MFMA slot (p * 4 NQ) / G
                        const int slot = q * 4 + s;
                        if (t_next >= 0) {
#pragma unroll
                            for (int p = 0; p < G; ++p)
                                if ((p * 4 * NQ) / G == slot) {
                                    if (p < OA::PER_WAVE) oa.dma_one(nb, m0, nk, wid, p);
                                    else ob.dma_one(nb + OA::FLOATS * 4, n0, nk, wid, p - OA::PER_WAVE);
                                }
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
            }
        } else {
            // 16x16x4: many narrow accumulators (TM x TN); the B fragments of two column tiles are fetched together and
            // their MFMAs alternate, so consecutive MFMAs never share an accumulator (40-cycle dependent latency
            // against a 32-cycle issue)
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                float fa[TM][4];
#pragma unroll
                for (int i = 0; i < TM; ++i) OA::template fetch<KG>(as, wm * WM + i * 16 + lr, q, lh, fa[i]);
#pragma unroll
                for (int j = 0; j < TN; j += 2) {
                    float fb0[4], fb1[4];
                    OB::template fetch<KG>(bs, wn * WN + j * 16 + lr, q, lh, fb0);
                    if (j + 1 < TN) OB::template fetch<KG>(bs, wn * WN + (j + 1) * 16 + lr, q, lh, fb1);
#pragma unroll
                    for (int s = 0; s < 4; ++s)
#pragma unroll
                        for (int i = 0; i < TM; ++i) {
                            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[i][s], fb0[s], acc[i][j], 0, 0, 0);
                            if (j + 1 < TN)
                                acc[i][j + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[i][s], fb1[s], acc[i][j + 1], 0, 0, 0);
                        }
                }
            }
        }
    };

    // ---- K loop: tile t lives in stage t % NST; NST - 1 tiles are in flight ahead of the one being multiplied.
    // Iteration t: (1) this wave's pieces of tile t have landed (counted vmcnt: the younger tiles stay in flight),
    // (2) barrier: every wave's pieces have, and every wave is done reading tile t - 1, (3) tile t + NST - 1 is
    // requested into the stage tile t - 1 occupied, (4) MFMAs of tile t.
#pragma unroll
    for (int s = 0; s < NST - 1; ++s)
        if (s < nfull) issue(s);
    for (int t = 0; t < nfull; ++t) {
        if (UNIFORM && NST > 2 && t + NST - 2 < nfull) wait_vmcnt<UNIFORM ? (NST - 2) * G : 0>();
        else wait_vmcnt<0>();
        asm volatile("" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if constexpr (SPREAD) {
            compute(smem + (t % NST) * STAGE, t + NST - 1 < nfull ? t + NST - 1 : -1);
        } else {
            if (t + NST - 1 < nfull) issue(t + NST - 1);
            compute(smem + (t % NST) * STAGE, -1);
        }
    }
    if (tail) {
        // the end of K inside a tile: image-K operands masked through registers, image-M operands by the descriptor
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        float* st = smem;
        const long k0 = kbeg + (long)nfull * BK;
        if constexpr (AKC) oa.fill_tail(st, g.A, m0, k0, g.M, kend); else oa.dma(smem_byte, m0, k0, wid);
        if constexpr (BKC) ob.fill_tail(st + OA::FLOATS, g.B, n0, k0, g.N, kend); else ob.dma(smem_byte + OA::FLOATS * 4, n0, k0, wid);
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        compute(st, -1);
    }

    // ---- split-K: publish the partial tile, the last arriver sums all slices in slice order (see gemm.hip)
    constexpr int NV4 = TM * TN * AR / 4;           // float4s per thread
    if (split > 1 && g.out_mode != 2) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();               // (smem is reused for the hand-off flag below)
        float4* slab = reinterpret_cast<float4*>(g.slabs) +
                       ((size_t)(tile - g.t0) * split + slice) * (size_t)(BM * BN / 4);
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r4 = 0; r4 < AR / 4; ++r4) {
                    const f32x4 vv = {acc[i][j][4 * r4], acc[i][j][4 * r4 + 1], acc[i][j][4 * r4 + 2], acc[i][j][4 * r4 + 3]};
                    float4* dst = slab + ((i * TN + j) * (AR / 4) + r4) * 256 + threadIdx.x;
                    asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(dst), "v"(vv) : "memory");
                }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        int* flag = reinterpret_cast<int*>(smem);
        if (threadIdx.x == 0)
            *flag = __hip_atomic_fetch_add(g.counters + tile, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
        if (*flag != split - 1) return;
        if (threadIdx.x == 0) {
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            __hip_atomic_store(g.counters + tile, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        __syncthreads();
        const float4* base = reinterpret_cast<const float4*>(g.slabs) + (size_t)(tile - g.t0) * split * (size_t)(BM * BN / 4);
        // fixed order: deterministic.  Four slabs' loads in flight per accumulator quad (round 6; one load per round trip before: a tile
        // in S slices paid S dependent trips per quad), a last group of fewer padded with zeros -- x + 0 = x, the bits do not change
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r4 = 0; r4 < AR / 4; ++r4) {
                    float4 sum = make_float4(0.f, 0.f, 0.f, 0.f);
                    const float4* src = base + ((i * TN + j) * (AR / 4) + r4) * 256 + threadIdx.x;
                    for (int sl = 0; sl < split; sl += 4) {
                        float4 v[4];
#pragma unroll
                        for (int u = 0; u < 4; ++u)
                            v[u] = sl + u < split ? src[(size_t)(sl + u) * (BM * BN / 4)] : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                        for (int u = 0; u < 4; ++u) { sum.x += v[u].x; sum.y += v[u].y; sum.z += v[u].z; sum.w += v[u].w; }
                    }
                    acc[i][j][4 * r4] = sum.x; acc[i][j][4 * r4 + 1] = sum.y;
                    acc[i][j][4 * r4 + 2] = sum.z; acc[i][j][4 * r4 + 3] = sum.w;
                }
    }
    (void)NV4;

    // ---- epilogue.  C/D layouts: 32x32: col = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5);
    //                             16x16: col = lane & 15, row = 4 (lane >> 4) + r.
    const bool add_bias = g.bias != nullptr && (slice == 0 || g.out_mode != 2);
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const long n = n0 + wn * WN + j * T + lr;
            if (n >= g.N) continue;
            const float bv = add_bias ? g.bias[n] : 0.f;
            unsigned short tq[4] = {0, 0, 0, 0};
            float csum = 0.f;
#pragma unroll
            for (int r = 0; r < AR; ++r) {
                const long m = m0 + wm * WM + i * T + (T == 32 ? (r & 3) + 8 * (r >> 2) + 4 * lh : 4 * lh + r);
                if (m >= g.M) {
                    if constexpr (BF16 == 2) {
                        // a ragged last group of four rows: its valid members go out one by one
                        if (g.CbT && (r & 3) > 0 && m - (r & 3) < g.M) {
                            for (int u = 0; u < (r & 3); ++u)
                                if (m - (r & 3) + u < g.M) g.CbT[n * g.ldcbt + m - (r & 3) + u] = tq[u];
                        }
                        // odd M: column M of the transposed copy is the k-pad of the product that reads it (K is taken
                        // in pairs) -- zero it here, the buffer may be shared with larger batches
                        if (g.CbT && m == g.M && (g.M & 1) && m < g.ldcbt) g.CbT[n * g.ldcbt + m] = 0;
                    }
                    continue;
                }
                float v = g.alpha * acc[i][j][r] + bv;
                if (g.act == 1) v = v > 0.f ? v : 0.f;
                else if (g.act == 2) v = v > 0.f ? v : 0.01f * v;
                if (g.mask_mode) {
                    float mv;
                    if (BF16 == 2 && g.mask16) {
                        const unsigned short mb = g.mask16[m * g.ldmask16 + n];
                        mv = (mb & 0x7fffu) != 0 && !(mb & 0x8000u) ? 1.f : 0.f;       // bf16 > 0
                    } else mv = g.mask[m * g.ldmask + n];
                    if (g.mask_mode == 1) v = mv > 0.f ? v : 0.f;
                    else v = mv > 0.f ? v : 0.01f * v;
                }
                if (BF16 != 2 || g.C) {
                    float* c = g.C + m * g.ldc + n;
                    if (g.out_mode == 0) *c = v;
                    else if (g.out_mode == 1) { v += *c; *c = v; }
                    else atomicAdd(c, v);
                }
                csum += v;
                if constexpr (BF16 == 2) {
                    const unsigned short bits = __builtin_bit_cast(unsigned short, (__bf16)v);
                    if (g.Cb) g.Cb[m * g.ldcb + n] = bits;                 // 32 lanes x 2 B: 64-byte runs along n
                    if (g.CbT) {
                        // the lane's registers r = 4 q .. 4 q + 3 are four CONSECUTIVE rows m: one 8-byte store per group
                        tq[r & 3] = bits;
                        if ((r & 3) == 3) {
                            unsigned short* dst = g.CbT + n * g.ldcbt + (m - 3);
                            if (m < g.M) {                                  // (m - 3 .. m all valid: m ascends inside a group)
                                uint2 pk;
                                pk.x = (unsigned)tq[0] | ((unsigned)tq[1] << 16);
                                pk.y = (unsigned)tq[2] | ((unsigned)tq[3] << 16);
                                *reinterpret_cast<uint2*>(dst) = pk;
                            }
                        }
                    }
                }
            }
            if constexpr (T == 32) {
                if (g.colsum) {
                    csum += __shfl_xor(csum, 32, 64);              // lanes l and l + 32: the same column, the other rows
                    if (lh == 0) g.colsum[((long)tm * (BM / 32) + wm * TM + i) * g.ldcs + n] = csum;
                }
            }
        }
}

template <int BM, int BN, int WM, int WN, int T, bool AKC, bool BKC, int NST, bool SPREAD = false, int BF16 = 0>
__global__ __launch_bounds__(256, 2) void gemm_glds_kernel(Args g) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    gemm_glds_body<BM, BN, WM, WN, T, AKC, BKC, NST, SPREAD, BF16>(g, (int)blockIdx.x, smem);
}

// Several independent problems of ONE layout in one launch (round 3: the parameter-gradient GEMMs of the whole MLP
// backward, dW_l = dY_l^T X_l -- four launches of 32 ... 256 tiles each, every one paying its own ~8 us of pipeline fill,
// output burst and launch latency at one block per CU, become one launch of 592 tiles).  Problem i owns blocks
// [blk0[i], blk0[i + 1]); inside them the block id means what it means in a launch of that problem alone (whole tiles
// / K slices / slabs and tickets at the problem's own scratch offsets).
constexpr int MAX_GROUP = 4;
struct GroupArgs {
    Args p[MAX_GROUP];
    int blk0[MAX_GROUP + 1];
};
template <int BM, int BN, int WM, int WN, int T, bool AKC, bool BKC, int NST, bool SPREAD = false, int BF16 = 0>
__global__ __launch_bounds__(256, 2) void gemm_glds_grouped_kernel(GroupArgs ga) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int bid = (int)blockIdx.x;
    int i = 0;
#pragma unroll
    for (int q = 1; q < MAX_GROUP; ++q) i += bid >= ga.blk0[q] ? 1 : 0;       // (blk0 of unused problems = total)
    gemm_glds_body<BM, BN, WM, WN, T, AKC, BKC, NST, SPREAD, BF16>(ga.p[i], bid - ga.blk0[i], smem);
}

template <int BM, int BN, int WM, int WN, int T, bool AKC, bool BKC, int NST, bool SPREAD = false, int BF16 = 0>
hipError_t launch_grouped(const GroupArgs& ga, hipStream_t s) {
    constexpr size_t lds = (size_t)NST * (BM + BN) * BK * sizeof(float);
    static NemoAttrOnce attr_once;
    auto kern = &gemm_glds_grouped_kernel<BM, BN, WM, WN, T, AKC, BKC, NST, SPREAD, BF16>;
    if (attr_once.need()) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(kern, dim3(ga.blk0[MAX_GROUP]), dim3(256), lds, s, ga);
    return hipSuccess;
}

template <int BM, int BN, int WM, int WN, int T, bool AKC, bool BKC, int NST, bool SPREAD = false, int BF16 = 0>
hipError_t launch(const Args& g, int blocks, hipStream_t s) {
    constexpr size_t lds = (size_t)NST * (BM + BN) * BK * sizeof(float);
    static NemoAttrOnce attr_once;
    auto kern = &gemm_glds_kernel<BM, BN, WM, WN, T, AKC, BKC, NST, SPREAD, BF16>;
    if (attr_once.need()) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), lds, s, g);
    return hipSuccess;
}

// Extents for the buffer descriptors.  Rows may be read up to the next multiple of four elements (a row's storage
// is 16-byte granular whenever ld % 4 == 0, which the DMA path requires anyway).
inline bool extents(int ta, int tb, long M, long N, long K, long lda, long ldb, unsigned* a_bytes, unsigned* b_bytes) {
    auto up4 = [](long x) { return (x + 3) / 4 * 4; };
    const long a_el = ta ? (K - 1) * lda + (up4(M) < lda ? up4(M) : lda) : (M - 1) * lda + (up4(K) < lda ? up4(K) : lda);
    const long b_el = tb ? (N - 1) * ldb + (up4(K) < ldb ? up4(K) : ldb) : (K - 1) * ldb + (up4(N) < ldb ? up4(N) : ldb);
    // every offset the kernel forms (tiles run up to 255 rows / 63 k past the end) must stay below 2^32
    const long a_max = ta ? (K + 64) * lda + M + 256 : (M + 256) * lda + K + 64;
    const long b_max = tb ? (N + 256) * ldb + K + 64 : (K + 64) * ldb + N + 256;
    if (a_max * 4 >= (1L << 32) || b_max * 4 >= (1L << 32) || K < 1) return false;
    *a_bytes = (unsigned)(a_el * 4);
    *b_bytes = (unsigned)(b_el * 4);
    return true;
}

}  // namespace glds
