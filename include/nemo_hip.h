/* libnemo_hip.so -- C ABI of the MI355X (gfx950) NeMo fitting kernels.
 *
 * Drop-in boundary for the per-iteration hot path of wangkua1/nemo-cvpr2023
 * (scripts/learned_multi_view_recon_nn.py -> nemo/neural_motion_model.py NemoV*.step).
 * The reference has no native/FFI layer on this path (everything is PyTorch aten ops), so
 * every entry point below cites the reference *operator* it replaces (file:line under the
 * reference tree).  The Python host in nemo_cvpr2023_amd/ binds these with ctypes
 * (INTEGRATION.md shows the stub a reference maintainer would add).
 *
 * Conventions
 *  - every function returns int32: 0 ok, <0 invalid argument, >0 a hipError_t;
 *  - all pointers are DEVICE pointers to contiguous row-major fp32 unless stated
 *    (index arrays are int64); the caller (PyTorch allocator) owns all memory; no
 *    allocation (nemo_ctx_create's immutable per-device constants excepted), no process-global
 *    state, re-entrant; the ONE piece of state outside the arguments is the calling host
 *    thread's binding of a caller-owned reduction arena (nemo_reduce_ws_bind below);
 *    "ld*" arguments are row strides in elements;
 *  - the last argument is the hipStream_t to launch on (as void*);
 *  - immutable model constants live in an opaque nemo_ctx created per device.
 *  - "_bwd" entry points *accumulate* (+=) into parameter-gradient outputs (the caller
 *    zeroes one flat gradient buffer per step) and *overwrite* activation gradients,
 *    unless stated.
 */
#ifndef NEMO_HIP_H
#define NEMO_HIP_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define NEMO_ABI_VERSION 18
int32_t nemo_abi_version(void);

/* Deterministic accumulation (round 5; caller-owned since ABI 17).  Every sum over the blocks of a launch that used float atomics
 * until round 4 -- the per-view loss / camera-gradient sums of the key-point kernels (nemo/neural_motion_model.py:3551-3558,
 * :3073-3124 under autograd), the phase-network gradients, bias column sums, the scalar losses -- is added in a FIXED order by the
 * launch's last-arriving block from per-block deposits in a scratch arena: the same inputs give the same bits, run to run.
 *   - The arena is CALLER-OWNED device memory: ws 256-byte aligned, bytes >= nemo_reduce_ws_bytes(N, V) for passes of up to N samples
 *     over V views, ZERO-FILLED ONCE when allocated (its head holds arrival tickets the kernels return to zero), on the device the
 *     launches run on.
 *   - nemo_reduce_ws_bind(ws, bytes) binds it to the CALLING HOST THREAD (thread-local; NULL unbinds) and rewinds its bump cursor;
 *     every launch of this library that reduces across blocks then takes its own region of the bound arena.
 *     nemo_reduce_scratch_reset() rewinds the cursor of the calling thread's arena.  Bind (or reset) at the top of every pass: the
 *     launches of a pass then own distinct regions, and a captured HIP graph keeps the regions it was captured with.
 *   - Threading / streams: launches that share an arena must be ordered on ONE stream (or by events); concurrent passes -- other
 *     streams, other devices, other host threads -- bind DIFFERENT arenas.  The library keeps no other state, no process globals.
 *   - A launch that finds no arena bound, or the arena exhausted, falls back to the float atomics of rounds 1 - 4 (correct, not
 *     bit-reproducible) and nemo_reduce_fallbacks() -- a process-wide count of such launches -- goes up: a determinism test asserts
 *     it stays 0.  NEMO_ORDERED_REDUCE=0 forces the atomics (not counted). */
int64_t nemo_reduce_ws_bytes(int64_t n_samples, int64_t n_views);
int32_t nemo_reduce_ws_bind(void* ws, int64_t bytes);
int32_t nemo_reduce_scratch_reset(void);
int64_t nemo_reduce_fallbacks(void);

/* ------------------------------------------------------------------------------------------
 * Dense fp32 contraction on the matrix cores (v_mfma_f32_32x32x2_f32), fused epilogue:
 *   C (op)= maskfn( act( alpha * opA(A) @ opB(B) + bias[n] ) )
 * transA=0: A[m*lda+k], 1: A[k*lda+m].  transB=0: B[k*ldb+n], 1: B[n*ldb+k].
 * act: 0 none, 1 ReLU, 2 LeakyReLU(0.01).  mask_mode: 0 none, 1 v*=(mask>0),
 * 2 v*=(mask>0 ? 1 : 0.01)  (activation backward from the saved *output*).
 * out_mode: 0 store, 1 C+=, 2 atomicAdd (act/mask must be 0).
 * split_k: 0 = chosen by the library (tile shape and K split from a cost model); n >= 1 = exactly n
 * K slices.  With out_mode 0/1 the slices of a tile are combined inside the launch by its
 * last-arriving block, in slice order (deterministic), and the epilogue stays fused; this needs the
 * caller-owned scratch `ws` (16-byte aligned, ws_bytes >= NEMO_GEMM_WS_MIN, ZERO-FILLED ONCE when
 * allocated -- its first 16 KB are arrival tickets the kernel returns to zero -- and not shared by
 * launches that may run concurrently).  ws == NULL: no in-launch combine (split_k > 1 then requires
 * out_mode 2).
 * Replaces nn.Linear fwd/bwd (nemo/neural_motion_model.py:58-71,130-148;
 * human_body_prior/models/vposer_model.py:69-88) and the pose-blend matmul
 * (human_body_prior/body_model/lbs.py:229-233).
 */
int32_t nemo_gemm_f32(int32_t transA, int32_t transB, int64_t M, int64_t N, int64_t K,
                      const float* A, int64_t lda, const float* B, int64_t ldb,
                      float* C, int64_t ldc, const float* bias, int32_t act,
                      const float* mask, int64_t ldmask, int32_t mask_mode, float alpha,
                      int32_t out_mode, int32_t split_k, void* ws, int64_t ws_bytes, void* stream);
#define NEMO_GEMM_WS_MIN (16384 + 65536)
/* nemo_gemm_f32 (out_mode 0, library-chosen plan) that ALSO leaves the column sums of its result -- after bias / activation /
 * mask -- per 32-row band: colsum[band * ldcs + n], band = 0 .. nemo_gemm_colsum_rows(M) - 1, ldcs >= N (round 5).  The
 * activation-gradient launches of the fp32 MotionNet backward (dX_l = mask(dY_{l+1} W_{l+1}), neural_motion_model.py:58-71 under
 * autograd) hand the next layer's bias gradient over this way: a sum over 2 ceil(M / 64) short rows instead of a second pass
 * over the M x N matrix (6.7 GB per step at 256 x 1024).  One writer per element: deterministic. */
int32_t nemo_gemm_f32_colsum(int32_t transA, int32_t transB, int64_t M, int64_t N, int64_t K,
                             const float* A, int64_t lda, const float* B, int64_t ldb,
                             float* C, int64_t ldc, const float* bias, int32_t act,
                             const float* mask, int64_t ldmask, int32_t mask_mode, float alpha,
                             float* colsum, int64_t ldcs, void* ws, int64_t ws_bytes, void* stream);
/* Same contraction, arguments and epilogues with both operands rounded to bf16 (RNE) on their way from LDS into the
 * matrix cores (v_mfma_f32_32x32x16_bf16), fp32 accumulate, fp32 in memory on both sides -- BASELINE configs[2].
 * Operands that do not qualify for the LDS-DMA kernel (rows not 16-byte aligned) are multiplied in fp32 instead. */
int32_t nemo_gemm_bf16(int32_t transA, int32_t transB, int64_t M, int64_t N, int64_t K,
                       const float* A, int64_t lda, const float* B, int64_t ldb,
                       float* C, int64_t ldc, const float* bias, int32_t act,
                       const float* mask, int64_t ldmask, int32_t mask_mode, float alpha,
                       int32_t out_mode, int32_t split_k, void* ws, int64_t ws_bytes, void* stream);
/* out[n] += sum_m X[m*ldx+n]   (bias gradients). */
/* bf16 operands IN MEMORY (round 3): C (M x N, fp32) (op)= maskfn(act(alpha * A @ B^T + bias)) with A (M x K) and B (N x K)
 * bf16 (uint16_t bit patterns), row-major, k-contiguous; K even, lda / ldb multiples of 8, 16-byte aligned bases.  Same
 * epilogues and scratch as nemo_gemm_f32; additionally the result can be stored as bf16 (Cb, row stride ldcb) and as its
 * bf16 TRANSPOSE (CbT (N x M), ldcbt % 4 == 0) -- the k-contiguous operands the following products of an MLP chain need
 * (forward: next layer's A; dX: next dY; dW = dY^T X: both operands transposed).  The values entering the matrix cores are
 * the ones nemo_gemm_bf16 rounds on the fly; half the bytes move.  C may be NULL when Cb / CbT are given (out_mode 0: only the
 * bf16 copies of the result are kept -- hidden activations).  mask_mode 17 / 18 = modes 1 / 2 with `mask` pointing at a
 * BF16 matrix (ldmask in bf16 elements), e.g. the bf16 copy of the ReLU output.  nemo_cast_bf16: dst (bf16) = src (fp32, rows x cols),
 * transposed when `transpose` != 0; the k-pad up to the next multiple of 8 (bounded by ldd) is zero-filled. */
int32_t nemo_gemm_bf16mem(int64_t M, int64_t N, int64_t K, const uint16_t* A, int64_t lda, const uint16_t* B, int64_t ldb,
                          float* C, int64_t ldc, const float* bias, int32_t act, const float* mask, int64_t ldmask,
                          int32_t mask_mode, float alpha, int32_t out_mode, uint16_t* Cb, int64_t ldcb, uint16_t* CbT,
                          int64_t ldcbt, float* colsum, int64_t ldcs, void* ws, int64_t ws_bytes, void* stream);
/* colsum != NULL (out_mode 0 / 1): the epilogue also leaves the column sums of the result per 32-row band in
 * colsum[band * ldcs + n], band = 0 .. nemo_gemm_colsum_rows(M) - 1, ldcs >= N -- the bias gradient sum_m dY[m][n] of a
 * layer (nemo/neural_motion_model.py:58-71 under autograd) is then nemo_colsum_multi over that short matrix, and dY need
 * not exist in fp32.  Every element has one writer: deterministic. */
int64_t nemo_gemm_colsum_rows(int64_t M);
/* nemo_gemm_f32 (fp32 operands, fp32 arithmetic, out_mode 0, no mask) whose result is stored as bf16 (Cb, ldcb >= N) and /
 * or as its bf16 transpose (CbT (N x M), ldcbt >= M, ldcbt % 4 == 0, 8-byte aligned) instead of / beside C (may be NULL):
 * the first MotionNet layer (nn.Linear(105, h), nemo/neural_motion_model.py:58-71) of the bf16-in-memory chain, whose
 * rows of 105 floats keep it on the fp32 path -- its output feeds bf16 products only. */
int32_t nemo_gemm_f32_b16out(int32_t transA, int32_t transB, int64_t M, int64_t N, int64_t K, const float* A, int64_t lda,
                             const float* B, int64_t ldb, float* C, int64_t ldc, const float* bias, int32_t act,
                             uint16_t* Cb, int64_t ldcb, uint16_t* CbT, int64_t ldcbt, void* ws, int64_t ws_bytes,
                             void* stream);
int32_t nemo_cast_bf16(int64_t rows, int64_t cols, const float* src, int64_t lds, uint16_t* dst, int64_t ldd,
                       int32_t transpose, void* stream);
/* fp32 -> THREE bf16 pieces per row for a split-precision product on the bf16 matrix cores (round 5: the first MotionNet layer,
 * nn.Linear(105, h) of nemo/neural_motion_model.py:58-71, under gemm_dtype = 'bf16'): with hi = bf16(x), lo = bf16(x - hi) and
 * seg = cols rounded up to 8, row r of dst (ldd >= 3 seg) is [hi | lo | hi] (order 0: the activation operand) or
 * [hi | hi | lo] (order 1: the weight operand), pads zero -- one NT product with K = 3 seg then sums hi*hi + lo*hi + hi*lo,
 * i.e. the fp32 product up to the 2^-17 terms, through nemo_gemm_bf16mem. */
int32_t nemo_cast_bf16_split3(int64_t rows, int64_t cols, const float* src, int64_t lds, uint16_t* dst, int64_t ldd,
                              int32_t order, void* stream);
/* ---- fp32-EQUIVALENT split precision on the 16-bit matrix cores for the MotionNet chain (round 6; csrc/gemm_xp.h) ----
 * Replaces, in fp32 builds (args.mlp_gemm = 'f32_split'), the nn.Linear products of MotionNet and their autograd
 * (nemo/neural_motion_model.py:58-71, :130-148): Y = X W^T, dX = dY W, dW = dY^T X, each as C = A B^T over k-contiguous
 * "xp matrices".  An xp matrix (rows x K, format fmt) holds every fp32 value x as NP = fmt 16-bit pieces whose sum is x:
 *   fmt 3: three bf16 pieces x0 = bf16(x), x1 = bf16(x - x0), x2 = bf16(x - x0 - x1) (= x exactly; no scale, fp32's range);
 *   fmt 2: two fp16 pieces of s x, x0 = fp16(s x), x1 = fp16(s x - x0), s a power of two chosen by the caller with |s x| < 65504;
 * piece p of element (r, k) is the 16-bit word r * ld + (k / 32) * 32 NP + p * 32 + k % 32, ld >= nemo_xp_ld(fmt, K) =
 * 32 NP ceil(K / 32), ld % 8 == 0, base 16-byte aligned, and elements k in [K, 32 ceil(K / 32)) of every row are ZERO.
 * nemo_gemm_xp: C (M x N, fp32, may be NULL) (op)= v, v = maskfn(act(alpha * A B^T + bias)), keeping the piece products of weight
 * >= 2^-24 (fmt 3: six, fmt 2: three; each exact in fp32, fp32 accumulation): error against float64 <= that of nemo_gemm_f32.
 * mask_mode 1: v = x0(maskx[m][n]) > 0 ? v : 0 with maskx an xp matrix (M x N) -- the copy of the ReLU output; 2 (ABI 18): ... : 0.01 v,
 * LeakyReLU' of the copy of a LeakyReLU output (VPoser's encoder, vposer_model.py:69-76, under the KL term's autograd).  Cx / CxT (may be
 * NULL): v * out_scale and its transpose written as xp matrices (M x N, ld ldcx / N x M, ld ldcxt) -- the operands of the next
 * products of the chain; colsum as nemo_gemm_bf16mem.  ws as nemo_gemm_f32.
 * nemo_cast_xp: up to NEMO_CAST_XP_MAX fp32 matrices (rows x cols, row stride lds) -> their xp copies dst (rows x cols) and / or
 * dstT (cols x rows) of `scale` * src in ONE launch (fmt 3 ignores scale).
 * fmt 2 scale records (device memory, float[NEMO_XP_META_FLOATS], 16-byte aligned: [0] the power-of-two scale s of the pieces, [2, 34)
 * absmax slots -- the absmax of the true values is their maximum; ZERO the record before its producer runs): the range guard of the
 * fp16 pieces lives on the device, inside captured graphs.  nemo_absmax_multi accumulates max |src| into the slots (desc.overwrite:
 * 32 blocks each STORE their partial maximum into their own slot -- the record needs no zeroing, the previous contents are gone);
 * nemo_cast_xp with desc.meta reads the absmax, scales by s = 2^floor(log2(2^15 / absmax)) (|s x| < 2^15 < 65504: cannot overflow) and
 * writes meta[0] = s; nemo_gemm_xp divides alpha by the operands' s (metaA[0], metaB[0]; NULL: 1) and, with metaOut, scales the
 * result's copies by s_out = 2^floor(log2(2^15 / (|alpha| K absmax_A absmax_B + absmax_bias))) -- a bound on |v|, so the copies cannot
 * overflow either --, writes metaOut[0] = s_out and accumulates the result's true absmax into metaOut's slots; metaZero (may be NULL): a
 * record whose absmax slots the launch returns to zero (the caller orders it behind their last reader) -- how a record that a
 * producer OUTSIDE the step's zero-filled arena accumulates into (X, by nemo_phase_embed_fwd_begin) is ready for the next pass.
 * What a loose bound costs is relative precision of entries below 2^-18 of it (fp16 subnormals), never range. */
#define NEMO_XP_META_FLOATS 64
#define NEMO_CAST_XP_MAX 8
typedef struct {
    const float* src; int64_t rows, cols, lds;
    uint16_t* dst; int64_t ldd;
    uint16_t* dstT; int64_t lddT;
    float scale;
    float* meta;        /* fmt 2, may be NULL: scale record (absmax in, scale out); overrides `scale` */
} nemo_cast_xp_desc;
typedef struct { const float* src; int64_t rows, cols, lds; float* meta; int32_t overwrite; } nemo_absmax_desc;
int64_t nemo_xp_ld(int32_t fmt, int64_t k);
int32_t nemo_gemm_xp(int32_t fmt, int64_t M, int64_t N, int64_t K, const uint16_t* A, int64_t lda, const uint16_t* B, int64_t ldb,
                     float* C, int64_t ldc, const float* bias, int32_t act, const uint16_t* maskx, int64_t ldmask,
                     int32_t mask_mode, float alpha, int32_t out_mode, uint16_t* Cx, int64_t ldcx, uint16_t* CxT, int64_t ldcxt,
                     float out_scale, float* colsum, int64_t ldcs, const float* metaA, const float* metaB, const float* metaBias,
                     float* metaOut, float* metaZero, void* ws, int64_t ws_bytes, void* stream);
int32_t nemo_cast_xp(int32_t fmt, int32_t n, const nemo_cast_xp_desc* descs, void* stream);
/* Up to NEMO_GEMM_GROUP_MAX products C_i (M_i x N_i, fp32) (op)= alpha_i A_i B_i^T over xp matrices in ONE launch (out_mode 0 store /
 * 1 +=; no bias / activation / mask / copies; metaA / metaB as nemo_gemm_xp): the four parameter gradients dW_l = dY_l^T X_l of the
 * MotionNet backward (nemo/neural_motion_model.py:58-71, :130-148 under autograd) behind its dX chain -- a nemo_gemm_xp launch takes whole
 * CUs, so dX and dW launches side by side only time-share; one grouped launch covers the chip ~2.7 times.  ws as nemo_gemm_f32, shared by
 * the problems (zero-filled once; >= 64 MB covers the chain at any batch size). */
typedef struct {
    int64_t M, N, K;
    const uint16_t* A; int64_t lda; const uint16_t* B; int64_t ldb; float* C; int64_t ldc;
    float alpha; int32_t out_mode;
    const float* metaA; const float* metaB;
} nemo_gemm_xp_problem;
int32_t nemo_gemm_xp_grouped(int32_t fmt, int32_t n, const nemo_gemm_xp_problem* problems, void* ws, int64_t ws_bytes, void* stream);
int32_t nemo_absmax_multi(int32_t n, const nemo_absmax_desc* descs, void* stream);
/* Up to NEMO_GEMM_GROUP_MAX independent products C_i (op)= alpha_i * opA(A_i) @ opB(B_i) (out_mode 0 store / 1 C +=; no
 * bias / activation / mask) in ONE launch when they share a layout and their operands are 16-byte aligned -- the
 * parameter gradients dW_l = dY_l^T X_l of the whole MotionNet backward (nemo/neural_motion_model.py:58-71,130-148 under
 * autograd): four launches of 32 ... 256 tiles, each paying its own pipeline fill and output burst at one block per CU,
 * become one of 592; the small problems are cut along K to fill the launch's last slots.  Anything else runs as
 * consecutive nemo_gemm_f32 calls with the same results.  `ws` as for nemo_gemm_f32 (the problems share it).
 * nemo_gemm_grouped_bf16: operands rounded to bf16 on their way into the matrix cores, as nemo_gemm_bf16. */
#define NEMO_GEMM_GROUP_MAX 4
typedef struct {
    int32_t transA, transB; int64_t M, N, K;
    const float* A; int64_t lda; const float* B; int64_t ldb; float* C; int64_t ldc;
    float alpha; int32_t out_mode;
} nemo_gemm_problem;
int32_t nemo_gemm_grouped_f32(int32_t n, const nemo_gemm_problem* problems /* HOST array */, void* ws, int64_t ws_bytes,
                              void* stream);
int32_t nemo_gemm_grouped_bf16(int32_t n, const nemo_gemm_problem* problems /* HOST array */, void* ws, int64_t ws_bytes,
                               void* stream);
int32_t nemo_colsum_f32(const float* X, int64_t M, int64_t N, int64_t ldx, float* out, void* stream);
/* The same for up to NEMO_COLSUM_MAX matrices in ONE launch (all bias gradients of the MLP backward). */
#define NEMO_COLSUM_MAX 8
typedef struct { const float* X; int64_t M, N, ldx; float* out; } nemo_colsum_desc;
int32_t nemo_colsum_multi(int32_t n, const nemo_colsum_desc* descs /* HOST array */, void* stream);

/* ------------------------------------------------------------------------------------------
 * Phase warp + RBF embedding + instance code -> MLP input rows.
 * K1/K2 of SURVEY.md: monotonic_network.py:23-39, nemo/rbf.py:47-74,
 * nemo/neural_motion_model.py:3647-3657,3740-3744,3755-3764.
 * X is (N+1, ldx): row s<N = [ phi_d(phase_s) (D cols, or the phase itself if D==0) | code[view_s] (C cols) ],
 * row N = the "phase 0, zero code" row used for trans_0.  Only the owning view's network is
 * evaluated per sample.  raw_phase (N) optional: overrides linspace(0,1,T)[frame_idx].
 * kernel_id: 0 quadratic 1 linear 2 gaussian 3 inverse_quadratic 4 multiquadric
 *            5 inverse_multiquadric 6 spline 7 poisson_one 8 poisson_two 9 matern32 10 matern52.
 * phase_out (N) receives the warped phase (saved for backward); den_out (N, optional) the denominator
 * o - z + 1e-6 of monotonic_network.py:39 (what nemo_phase_embed_bwd takes as `ws`).  shifts/scales row v starts at
 * shifts + v*ldp / scales + v*ldp (ldp = 2K when the V networks are stored [sh_0|sc_0|sh_1|...]).
 * code_noise (N,C) optional: additive instance-code noise of NemoV3/V4 (:3921-3923).
 */
int32_t nemo_phase_embed_fwd(int64_t N, int64_t V, int64_t T, int64_t K, int64_t D, int64_t C,
                             const int64_t* view_idx, const int64_t* frame_idx, const float* raw_phase,
                             const float* shifts, const float* scales, int64_t ldp,
                             const float* log_sigmas, const float* codes, const float* code_noise,
                             int32_t kernel_id, float* X, int64_t ldx, float* phase_out, float* den_out,
                             float* x_meta, void* stream);
/* x_meta (may be NULL; ABI 17): the scale record of X (nemo_gemm_xp fmt 2) -- the launch accumulates max |X| into its absmax slots,
 * so the split-precision MotionNet chain needs no pass over X of its own. */
/* dX (N+1, ldx) -> d_shifts,d_scales (V rows of stride ldp), d_log_sigmas (D), d_codes (V,C); all
 * accumulated.  One launch: block (view, slice of 64 nodes) sums the phase-network gradient of its nodes over ALL samples of
 * its view in sample order (no cross-block sum, no atomics: deterministic, round 5); further blocks reduce the log_sigma /
 * code columns.  sorted_by_view != 0: the caller vouches that view_idx is non-decreasing (every full batch) -- a view's
 * samples are then found by a search instead of a scan of the batch; 0 is always correct.
 * ws (N floats, optional): the `den_out` the forward call wrote for the SAME inputs (the backward then skips two
 * K-long sigmoid sums per view); NULL: they are re-evaluated here. */
int32_t nemo_phase_embed_bwd(int64_t N, int64_t V, int64_t T, int64_t K, int64_t D, int64_t C,
                             const int64_t* view_idx, const int64_t* frame_idx, const float* raw_phase,
                             const float* shifts, const float* scales, int64_t ldp,
                             const float* log_sigmas, int32_t kernel_id, const float* phase,
                             const float* dX, int64_t ldx, float* ws,
                             float* d_shifts, float* d_scales, float* d_log_sigmas, float* d_codes,
                             int32_t sorted_by_view, void* stream);
/* nemo_phase_embed_bwd and nemo_colsum_multi(n_cs, descs) (same arguments, same results) in ONE launch: the batched bias
 * column sums of the MLP backward (nemo/neural_motion_model.py:58-71 under autograd) need only the activation gradients the
 * dX chain has produced by the time the phase backward starts, so they run in further blocks of its grid instead of as a
 * launch of their own behind the last parameter-gradient GEMM. */
int32_t nemo_phase_embed_bwd_colsum(int64_t N, int64_t V, int64_t T, int64_t K, int64_t D, int64_t C,
                                    const int64_t* view_idx, const int64_t* frame_idx, const float* raw_phase,
                                    const float* shifts, const float* scales, int64_t ldp, const float* log_sigmas,
                                    int32_t kernel_id, const float* phase, const float* dX, int64_t ldx, float* ws,
                                    float* d_shifts, float* d_scales, float* d_log_sigmas, float* d_codes, int32_t n_cs,
                                    const nemo_colsum_desc* descs /* HOST array */, int32_t sorted_by_view, void* stream);

/* ------------------------------------------------------------------------------------------
 * rot6d -> rotation matrix -> axis-angle, per (row, joint).
 * hmr/geometry.py:47-61 (rot6d_to_rotmat), :181-346 (rotation_matrix_to_angle_axis; 4-branch
 * quaternion, eps 1e-6; NaN->0 when zero_nan!=0), == human_body_prior/tools/rotation_tools.py:73-81
 * with zero_nan=0 (VPoser decoder, vposer_model.py:32-45,98-106).
 * rot6d (rows, ld6) holds J*6 numbers per row; R (rows,J,9) and aa (rows,J*3) may be NULL.
 */
int32_t nemo_rot6d_fwd(int64_t rows, int64_t J, const float* rot6d, int64_t ld6, int32_t zero_nan,
                       float* R, float* aa, void* stream);
/* d_rot6d (rows, ld6) = J^T [dR ; daa]  (either may be NULL); overwrites the J*6 columns. */
int32_t nemo_rot6d_bwd(int64_t rows, int64_t J, const float* rot6d, int64_t ld6, int32_t zero_nan,
                       const float* dR, const float* daa, float* d_rot6d, int64_t ldd, void* stream);
/* The step's backward of the pose head in ONE launch = nemo_v2v_prep_bwd (when dR2 != NULL; its results are
 * consumed in registers, dR / daa are NOT modified) + nemo_rot6d_bwd over N rows of 24 joints +
 * nemo_scale_neg_rowsum (when dTR != NULL: dTR[N][0..2] = - sum_{s<N} dTR[s][0..2], :3764-3766).
 * zero_row != 0: d_rot6d has N + 1 rows and the 144 rotation columns of row N -- the "phase 0 / zero code" row that
 * only yields trans_0 (:3755-3766) and so carries no rotation gradient -- are set to zero (a buffer shared by several
 * batch sizes otherwise keeps an earlier, larger batch's row there). */
int32_t nemo_pose_bwd_fused(int64_t N, const float* rot6d, int64_t ld6, int32_t zero_nan, const float* dR,
                            const float* daa, float* d_rot6d, int64_t ldd, const float* aa, const float* dR2,
                            float v2v_scale, float* dTR, int64_t ldt, int32_t zero_row, float* head_meta, void* stream);
/* head_meta (may be NULL; ABI 17): the scale record of the head gradient (nemo_gemm_xp fmt 2) -- the launch accumulates the absmax of the
 * 144 rotation columns it writes and, with dTR, of the three translation columns (row N included) into its slots. */
/* Stand-alone conversions (API parity with hmr/geometry.py). */
int32_t nemo_rotmat_to_aa(int64_t M, const float* R, int32_t zero_nan, float* aa, void* stream);
/* hmr/geometry.py:9-45: quaternion-form Rodrigues, angle = ||theta + 1e-8||; form=1 selects the
 * matrix form of human_body_prior/body_model/lbs.py:303-334 (eval path, forward only). */
int32_t nemo_rodrigues_fwd(int64_t M, const float* theta, int32_t form, float* R, void* stream);
int32_t nemo_rodrigues_bwd(int64_t M, const float* theta, const float* dR, float* dtheta, void* stream);

/* ------------------------------------------------------------------------------------------
 * SMPL model context (immutable constants on the device).
 * smplx.SMPL buffers + hmr/smpl.py:17-43 (J_regressor_extra, joint_map).  All inputs are HOST
 * pointers.  posedirs is (207, 3*NV) as smplx stores it; lbs_weights (NV,24); J_regressor (24,NV);
 * J_regressor_extra (n_extra,NV); sel_vertex_ids (n_sel) are the VertexJointSelector picks
 * (joints 24..24+n_sel-1 of the 45+n_extra superset); out_joints (n_out) index that superset and
 * select what nemo_kp_* produce (V0-3: joint_map[[38]+[1..24]], V4: joint_map[0..24]).
 * The context pre-contracts every non-kinematic output joint (a linear functional of the posed
 * mesh) with the skinning weights and pose blend-shapes so the 2-D objective never touches
 * the 6890-vertex mesh.
 */
typedef struct nemo_ctx nemo_ctx;
int32_t nemo_ctx_create(nemo_ctx** out, int64_t NV, const float* v_template, const float* shapedirs,
                        const float* posedirs, const float* J_regressor, const float* lbs_weights,
                        const int64_t* parents, int64_t n_extra, const float* J_regressor_extra,
                        int64_t n_sel, const int64_t* sel_vertex_ids,
                        int64_t n_out, const int64_t* out_joints);
/* betas: HOST (10); recomputes v_shaped, rest joints and the shape-dependent constants
 * (lbs.py:209-216).  Called once by nemo_ctx_create with zeros. */
int32_t nemo_ctx_set_betas(nemo_ctx* ctx, const float* betas);
int32_t nemo_ctx_destroy(nemo_ctx* ctx);
int64_t nemo_ctx_num_verts(const nemo_ctx* ctx);
/* Sparse skinning (round 4).  lbs.py:236-241 forms T = W A as a dense (V x 24) x (24 x 16) product; the published SMPL
 * model has at most four non-zero weights per vertex.  nemo_ctx_skin_nnz: the largest number of non-zero lbs_weights any
 * vertex has.  When it is <= 4, nemo_v2v_fused(_bf16 / _bf16mem) skins with the non-zero weights only (same sums without the
 * zero terms); nemo_ctx_set_skin_sparse(ctx, 0) selects the dense product again (EINVAL for enable != 0 when
 * skin_nnz > 4), nemo_ctx_skin_sparse reports the current choice.  Change it only while no launch is in flight. */
int32_t nemo_ctx_skin_nnz(const nemo_ctx* ctx);
/* Range guard of the fp16 split-precision mesh kernel (ABI 17).  nemo_v2v_fused_split(mem) stage the blended vertices
 * vp = v_shaped + P pf (human_body_prior/body_model/lbs.py:229-233) as two fp16 pieces of 2^12 vp.  nemo_ctx_vp_bound: the bound
 * max_v (|v_shaped[v]| + 2 sum_k |posedirs[k][v]|) >= |vp| for every pose (|pf| <= 2), recomputed by nemo_ctx_set_betas;
 * nemo_ctx_split_ok: 1 while 2^12 * bound < 2^15.9 (a body model in metres: bound ~ 1.5), else 0 -- then nemo_v2v_fused_split runs its
 * three-bf16-piece form (fp32's exponent range; same fp32-equivalent arithmetic, 14 % slower) and nemo_v2v_fused_splitmem returns
 * NEMO_EINVAL: the caller keeps d vp in fp32 (nemo_v2v_fused_split + the fp32 adjoint GEMM). */
int32_t nemo_ctx_split_ok(const nemo_ctx* ctx);
float nemo_ctx_vp_bound(const nemo_ctx* ctx);
/* nemo_ctx_skin_mfma_ok: 1 while, in addition, the relative joint transforms' entries (bounded from the rest joints: |J_0| + the longest
 * chain of bone lengths + max |J|) keep 2^12 |A| < 2^15.9 -- then nemo_v2v_fused_split(mem / xp) also run their two skinnings
 * (lbs.py:236-252) as fp16 split-precision MFMAs (kernel MODE 6: weights as three fp16 pieces = exactly, transforms as two, four piece
 * products: fp32-equivalent, error against float64 <= the fp32 kernel's), the dense 24-joint product on the 16-bit pipe instead of the
 * <= 4 non-zero weights on the VALU; NEMO_MESH_SKIN=sparse keeps the VALU form (MODE 5). */
int32_t nemo_ctx_skin_mfma_ok(const nemo_ctx* ctx);
int32_t nemo_ctx_skin_sparse(const nemo_ctx* ctx);
int32_t nemo_ctx_set_skin_sparse(nemo_ctx* ctx, int32_t enable);
int64_t nemo_ctx_nq(const nemo_ctx* ctx);            /* # non-kinematic output joints            */
const float* nemo_ctx_C1(const nemo_ctx* ctx);       /* device (207, nq*72) pre-contracted basis   */
const float* nemo_ctx_c0(const nemo_ctx* ctx);       /* device (nq*72) shape-dependent offset      */
const float* nemo_ctx_posedirs(const nemo_ctx* ctx); /* device (207, ld) zero-padded rows              */
int64_t nemo_ctx_posedirs_ld(const nemo_ctx* ctx);   /* ld = 3*NVp, NVp = NV rounded up to 16       */
const float* nemo_ctx_v_shaped(const nemo_ctx* ctx); /* device (3*NV)                              */

/* Forward kinematics (human_body_prior/body_model/lbs.py:350-404 batch_rigid_transform +
 * :229 pose_feature).  R (rows,24,9) -> A (rows,24,12) [3x4 relative transforms, row-major],
 * Jp (rows,24,3) posed joints, PF (rows, ldpf>=207) = (R[1:]-I) flattened (optional). */
int32_t nemo_fk_fwd(const nemo_ctx* ctx, int64_t rows, const float* R, float* A, float* Jp, float* PF,
                    int64_t ldpf, void* stream);
/* dA (rows,24,12), optional dJp (rows,24,3), dPF (rows, lddpf) -> dR (rows,24,9). */
int32_t nemo_fk_bwd(const nemo_ctx* ctx, int64_t rows, const float* R, const float* A, const float* dA,
                    const float* dJp, const float* dPF, int64_t lddpf, float* dR, void* stream);

/* ------------------------------------------------------------------------------------------
 * Keypoint objective: output joints -> +translation -> per-view camera -> 2-D loss.
 * nemo/neural_motion_model.py:3073-3124 (learned_camera_projection), hmr/geometry.py:78-106,
 * :2806-2843 (keypoint_loss), nemo/utils/misc_utils.py:91-105 (GMoF), :3551-3558 (per-view mean).
 * Mq (N, ldq>=nq*72) = PF @ C1 + c0 (one nemo_gemm_f32).  TR (N+1, ldt): predicted translation,
 * row N = trans_0 (pass add_trans=0 to skip).  cams (V,9) = [t(3) | rot6d(6)].  targets
 * (V,T,25,3) = (x, y, conf); gt_size (V,T).  loss_type: 0 mse_robust 1 mse 2 rmse 3 rmse_robust
 * 4 mse_robust_resized 5 rmse_resized.  Outputs (each may be NULL): j3d (N,n_out,3), p2d (N,n_out,2),
 * loss_all (N,n_out,W) with W=2 (mse*) or 1 (rmse*); view_acc (V,2) += [sum(loss*conf), #samples].
 *
 * PADDED BATCHES (n_valid, also on nemo_kp_bwd_ex / nemo_kl_fwd_bwd / nemo_gmm_fwd_bwd / nemo_pose3d_fwd_bwd /
 * nemo_v2v_prep_fwd): a DEVICE scalar or NULL.  Samples s >= *n_valid are padding (a rank's share of a random
 * minibatch rounded up to a fixed launch size, so that one captured HIP graph serves every share of that size,
 * scripts/learned_multi_view_recon_nn.py:291-296 sharded by instance): they carry valid indices, run through every
 * kernel, but contribute no loss, are not counted in any per-view / per-sample normaliser's sample count, and
 * receive exactly-zero gradients.  Kernels that divide by N keep dividing by the N they are launched with: the caller
 * folds N_launch / N_global into the weight it passes.
 */
int32_t nemo_kp_fwd(const nemo_ctx* ctx, int64_t N, int64_t V, int64_t T, const float* A, const float* Jp,
                    const float* Mq, int64_t ldq, const float* TR, int64_t ldt, int32_t add_trans,
                    const int64_t* view_idx, const int64_t* frame_idx, const float* cams,
                    const float* targets, const float* gt_size, float focal, float cx, float cy,
                    int32_t loss_type, int32_t mean_mode, float* j3d, float* p2d, float* loss_all,
                    float* view_acc, const int64_t* n_valid, void* stream);
/* mean_mode 0 (step, :3551-3558): *scalar_out += (1/n_U) sum_{v present} view_acc[v,0] /
 * (view_acc[v,1]*n_out*W), norm[0] = n_U.  mean_mode 1 (camera_fitting_loss, :2845-2867): plain mean
 * over all N*n_out*W elements of loss_all (no confidence weight), norm[0] = N.  norm is a device
 * scalar consumed by nemo_kp_bwd.  The same mean_mode must be given to nemo_kp_fwd / _bwd. */
int32_t nemo_kp_finalize(int64_t V, int64_t n_out, int32_t W, int32_t mean_mode, const float* view_acc,
                         float* scalar_out, float* norm, void* stream);
/* upstream = d total / d kp_loss.  Outputs: dA (N,24,12) and dMq (N,ldq) overwritten; dJp (N,24,3)
 * accumulated (caller zeroes it); dTR (N, lddt) rows overwritten (row N, the trans_0 gradient, is
 * -sum of the rows: nemo_scale_neg_rowsum); d_cams (V,9) accumulated.  dA==NULL skips the body
 * gradients (camera-only fitting, :2869-2906).  norm==NULL: the normaliser is derived from view_acc
 * inside the launch (nemo_kp_finalize then only produces the loss scalar and can run concurrently). */
int32_t nemo_kp_bwd(const nemo_ctx* ctx, int64_t N, int64_t V, int64_t T, const float* A, const float* Jp,
                    const float* Mq, int64_t ldq, const float* TR, int64_t ldt, int32_t add_trans,
                    const int64_t* view_idx, const int64_t* frame_idx, const float* cams,
                    const float* targets, const float* gt_size, float focal, float cx, float cy,
                    int32_t loss_type, int32_t mean_mode, const float* view_acc, const float* norm,
                    float upstream, float* dA, float* dJp, float* dMq, float* dTR, int64_t lddt,
                    float* d_cams, void* stream);
/* nemo_kp_fwd + nemo_kp_bwd in ONE launch (round 4).  The backward needs one thing from a COMPLETED forward: the per-view
 * normaliser of :3551-3558 -- and that depends on the batch's indices alone: view_count[v] = number of (unpadded) samples of
 * view v in the batch (device int64[V]; the caller knows it from the indices it drew: a constant for a full batch).  With
 * it the launch evaluates projection and loss once, writes the forward's outputs (j3d, p2d, loss_all: each may be NULL;
 * view_acc (V,2) += [sum(loss*conf), #samples], from which nemo_kp_finalize makes the loss scalar) and the backward's
 * (as nemo_kp_bwd).  Same results as the two launches; view_count must equal what the forward would have counted. */
int32_t nemo_kp_fwd_bwd(const nemo_ctx* ctx, int64_t N, int64_t V, int64_t T, const float* A, const float* Jp,
                        const float* Mq, int64_t ldq, const float* TR, int64_t ldt, int32_t add_trans,
                        const int64_t* view_idx, const int64_t* frame_idx, const float* cams, const float* targets,
                        const float* gt_size, float focal, float cx, float cy, int32_t loss_type, int32_t mean_mode,
                        const int64_t* view_count, float upstream, float* j3d, float* p2d, float* loss_all, float* view_acc,
                        float* dA, float* dJp, float* dMq, float* dTR, int64_t lddt, float* d_cams,
                        const int64_t* n_valid, void* stream);
/* nemo_kp_bwd with an additional gradient dj3d_extra (N, n_out, 3) w.r.t. the 3-D output joints (world space,
 * translation included) added before the pull-back through FK / the mesh functionals; cameras do not see it. */
int32_t nemo_kp_bwd_ex(const nemo_ctx* ctx, int64_t N, int64_t V, int64_t T, const float* A, const float* Jp,
                       const float* Mq, int64_t ldq, const float* TR, int64_t ldt, int32_t add_trans,
                       const int64_t* view_idx, const int64_t* frame_idx, const float* cams,
                       const float* targets, const float* gt_size, float focal, float cx, float cy,
                       int32_t loss_type, int32_t mean_mode, const float* view_acc, const float* norm,
                       float upstream, float* dA, float* dJp, float* dMq, float* dTR, int64_t lddt,
                       float* d_cams, const float* dj3d_extra, const int64_t* n_valid, void* stream);
/* Temporal smoothness of the output joints (an OPTIONAL term, not part of the published NemoV* step; formula
 * of humor/humor/fitting/fitting_loss.py:366-370): j3d (V*T, J, 3) laid out (view, frame);
 * *scalar_out += 0.5 * sum_{v,t<T-1,j} |j3d[v,t+1,j] - j3d[v,t,j]|^2;  dj3d (same shape, optional) = weight *
 * its gradient (overwritten). */
int32_t nemo_smooth_fwd_bwd(int64_t V, int64_t T, int64_t J, const float* j3d, float weight, float* scalar_out,
                            float* dj3d, void* stream);
/* Projection only (API parity: learned_camera_projection on arbitrary points (N,Jn,3)). */
int32_t nemo_project(int64_t N, int64_t Jn, int64_t V, const float* pts, const int64_t* view_idx,
                     const float* cams, float focal, float cx, float cy, float* p2d, void* stream);

/* ------------------------------------------------------------------------------------------
 * Full-mesh skinning (lbs.py:238-252) on a pose-blended mesh VP = PF @ posedirs + v_shaped.
 * nemo_skin_vertices: verts (rows,NV,3) (+ trans (rows,3), may be NULL) -- get_preds()['v'].
 */
int32_t nemo_skin_vertices(const nemo_ctx* ctx, int64_t rows, const float* VP, int64_t ldvp,
                           const float* A, const float* trans, int64_t ldt, float* verts, void* stream);

/* VPoser v2v term (nemo/neural_motion_model.py:2787-2793): rows [0,N) of VP/A are the
 * "orig" bodies, rows [N,2N) the detached VPoser reconstructions.
 *   loss_sum += sum |v_rec - v_orig|            (caller divides by N*NV*3)
 *   dVP (N, lddvp) = d(sum)/d VP_orig,  dA (N,24,12) = d(sum)/d A_orig      (overwritten)
 * i.e. the gradient of the un-normalised L1 sum is produced in the same pass. */
int32_t nemo_v2v_skin_l1(const nemo_ctx* ctx, int64_t N, const float* VP, int64_t ldvp, const float* A,
                         float* loss_sum, float* dVP, int64_t lddvp, float* dA, void* stream);
/* Same term, fused: pose blend + skinning of both bodies + L1 + gradient in one MFMA kernel; the
 * blended mesh never reaches HBM.  PF2 (2N, ldpf) pose features [orig rows | reconstruction rows] with
 * column 207 (if ldpf > 207) ZERO, A2 (2N,24,12).  loss_sum += sum|v_rec - v_orig|;
 * dVPt (3*NVp rows, NVp = NV rounded up to 16; ldn >= N rounded up to 16) = TRANSPOSED
 * d(sum)/dVP_orig (operand of the blend-shape adjoint GEMM with transA=1; pad rows/columns are
 * written with zeros); dA (N,24,12) = d(sum)/dA_orig (OVERWRITTEN; summed over the vertex ranges in
 * a fixed order, deterministic).  loss_sum is accumulated by ONE thread of the launch (the last block to
 * arrive adds the per-block partials in block order, in float64): bit-reproducible run to run.
 * ws: caller-owned scratch of nemo_v2v_fused_ws_bytes(ctx, N) bytes, 16-byte aligned, zero-filled once at
 * allocation (arrival tickets the kernel returns to zero), not shared by concurrent launches.  The tickets sit at
 * fixed offsets, so launches of DIFFERENT N may share one buffer of the largest size (the chunks of a big batch).
 * N <= 16 * 65536 per launch. */
int64_t nemo_v2v_fused_ws_bytes(const nemo_ctx* ctx, int64_t N);
int32_t nemo_v2v_fused(const nemo_ctx* ctx, int64_t N, const float* PF2, int64_t ldpf, const float* A2,
                       float* loss_sum, float* dVPt, int64_t ldn, float* dA, void* ws, int64_t ws_bytes,
                       void* stream);
/* nemo_v2v_fused in fp32-EQUIVALENT arithmetic with the pose blend's products (lbs.py:229-233, K = 207: 76 % of the fp32
 * kernel's matrix-pipe cycles) and the vertex->joint adjoint dA = W^T dT on the 16-bit matrix cores: every operand is carried as
 * TWO fp16 pieces of s x (s a power of two that keeps them in fp16's normal range) -- x0 = fp16(s x), x1 = fp16(s x - x0):
 * 11 + 11 significant bits and the remainder's sign = the fp32 value to one ulp (2^-23) in the worst case --, a product keeps
 * x0 y0 + x0 y1 + x1 y0 (each exact in fp32; the dropped x1 y1 <= 2^-22 |x y|), fp32 accumulation, the scales divided out exactly
 * afterwards.  Over the 207 terms of a blend / the 6890 of an adjoint sum the fp32 accumulation's own rounding dominates: measured
 * error against float64 = the fp32 product's (5.0e-7 against 5.1e-7 of the result's scale, tests/test_split_precision.py).  Blend shapes and
 * skinning weights are split once at nemo_ctx_create, pose features when staged, dT = +-[vp; 1] from the pieces of vp with the
 * sign bit flipped.  Skinning, L1 and d vp as in nemo_v2v_fused.  Same arguments, outputs, scratch and determinism.
 * tests/test_gpu_ops.py::test_v2v_fused_split_* hold its error against a float64 evaluation to the fp32-MFMA kernel's, and
 * tests/test_split_precision.py the arithmetic itself (round 5; the engine's `mesh_blend = 'f32_split'`; NEMO_MESH_PIECES=3: the
 * round's first form, three bf16 pieces per operand and six piece products). */
int32_t nemo_v2v_fused_split(const nemo_ctx* ctx, int64_t N, const float* PF2, int64_t ldpf, const float* A2,
                              float* loss_sum, float* dVPt, int64_t ldn, float* dA, void* ws, int64_t ws_bytes,
                              void* stream);
/* nemo_v2v_fused_split with d vp handed over as TWO fp16 piece planes of 2^12 d vp, NOT transposed (row = sample, 16 * ceil(N / 16)
 * rows of ldk >= the blend-shape row stride elements, plane 1 `plane` elements behind plane 0) -- the A operand of
 * nemo_gemm_f16x2mem_adj, which multiplies them with the two piece planes of s P in fp32-equivalent split precision
 * (A0 B0^T + A0 B1^T + A1 B0^T on v_mfma_*_f16, the 64 x 208 tile of the blend-shape adjoint, lbs.py:229-233 backward):
 * C (M x N) (op)= alpha * (...), 128 < N <= 208, K even, lda / ldb / plane strides multiples of 8 elements; alpha carries 1 / (the
 * pieces' scales); ws as nemo_gemm_f32 (round 5, ABI 16; the engine uses the pair from 256 samples on). */
int32_t nemo_v2v_fused_splitmem(const nemo_ctx* ctx, int64_t N, const float* PF2, int64_t ldpf, const float* A2, float* loss_sum,
                                uint16_t* dVPh, int64_t ldk, int64_t plane, float* dA, void* ws, int64_t ws_bytes, void* stream);
/* nemo_v2v_fused_splitmem with d vp handed over as ONE xp matrix (nemo_gemm_xp fmt 2: the two fp16 pieces of 2^12 d vp interleaved per
 * k-block of 32; 16 * ceil(N / 16) rows of ldk >= nemo_xp_ld(2, 3 * 16 ceil(NV / 16)) elements, ldk % 8 == 0) -- the A operand of the
 * blend-shape adjoint dPF = d vp P^T (lbs.py:229-233 backward) through nemo_gemm_xp against the xp copy of the blend shapes, metaA[0] =
 * 4096: one pass over d vp and the blend shapes per tile with all three piece products (round 6, ABI 17; 8 x 300: 150 -> ~90 us).
 * NEMO_EINVAL outside the range guard (nemo_ctx_split_ok == 0), as nemo_v2v_fused_splitmem. */
int32_t nemo_v2v_fused_splitxp(const nemo_ctx* ctx, int64_t N, const float* PF2, int64_t ldpf, const float* A2, float* loss_sum,
                               uint16_t* dVPx, int64_t ldk, float* dA, void* ws, int64_t ws_bytes, void* stream);
int32_t nemo_gemm_f16x2mem_adj(int64_t M, int64_t N, int64_t K, const uint16_t* A, int64_t lda, int64_t a_plane, const uint16_t* B,
                               int64_t ldb, int64_t b_plane, float* C, int64_t ldc, float alpha, int32_t out_mode, void* ws,
                               int64_t ws_bytes, void* stream);
/* The same with the pose blend (lbs.py:229-233, K = 207) on the bf16 matrix cores: blend shapes rounded to bf16 once
 * at nemo_ctx_create, pose features rounded when staged, fp32 accumulate; skinning, L1 and d vp in fp32; the vertex->joint adjoint dA on the
 * bf16 cores in split precision (two bf16 pieces per fp32 operand, 16 significant bits), and since round 5 the two skinning
 * products as well (NEMO_MESH_SPLIT=3: skinning in fp32, the round-3/4 kernel).
 * BASELINE configs[2] ("bf16"); not covered by the 1e-4 parity gate (tests state the bf16 tolerance). */
/* nemo_v2v_fused_bf16 with d vp written as bf16 and NOT transposed (round 3): dVPb has 16 * ceil(N / 16) rows (one per
 * sample) of ldk >= 3 * NVp bf16 -- the k-contiguous A operand of the blend-shape adjoint through nemo_gemm_bf16mem. */
int32_t nemo_v2v_fused_bf16mem(const nemo_ctx* ctx, int64_t N, const float* PF2, int64_t ldpf, const float* A2,
                               float* loss_sum, uint16_t* dVPb, int64_t ldk, float* dA, void* ws, int64_t ws_bytes,
                               void* stream);
/* dA == NULL in nemo_v2v_fused(_bf16): DEFERRED combine -- the launch leaves the per-block partial dA images in `ws`
 * and nemo_v2v_combine (same ctx, N, ws; any stream ordered behind the fused launch) sums them into dA (N,24,12),
 * overwriting.  The step runs it beside the blend-shape adjoint GEMM: the last-arriver reduction inside the fused
 * launch (three dependent memory round trips at the very end of a launch that fills the machine) leaves the critical
 * path.  Same summation order as the in-launch path: bit-identical dA. */
int32_t nemo_v2v_combine(const nemo_ctx* ctx, int64_t N, float* dA, const void* ws, int64_t ws_bytes, void* stream);
int32_t nemo_v2v_fused_bf16(const nemo_ctx* ctx, int64_t N, const float* PF2, int64_t ldpf, const float* A2,
                            float* loss_sum, float* dVPt, int64_t ldn, float* dA, void* ws, int64_t ws_bytes,
                            void* stream);
/* Builds the (2N,24,9) rotation set of the two bodies from the MLP pose:
 * rows<N: [R[:,0], Rodrigues(aa[:,3:72])], rows>=N: [R[:,0], Rodrigues(cat(aa_dec, aa[:,66:72]))]
 * (:2783-2791, hmr/geometry.py:9-45).  Padding samples (see nemo_kp_fwd) get the FIRST body twice: the full-mesh
 * L1 term and its gradient are then exactly zero for them. */
int32_t nemo_v2v_prep_fwd(int64_t N, const float* R, const float* aa, const float* aa_dec, float* R2,
                          const int64_t* n_valid, void* stream);
/* The same from the VPoser decoder's 6-D output dec6d (N,21,6; row stride lddec >= 126): the conversion 6-D -> rotation
 * matrix -> axis-angle (vposer_model.py:100-113, what nemo_rot6d_fwd(N, 21, ..., aa) computes) happens in the same launch and
 * aa_dec_out (N,63) receives the axis-angle form -- one launch less on the chain the mesh kernel waits for. */
int32_t nemo_v2v_prep_fwd_dec(int64_t N, const float* R, const float* aa, const float* dec6d, int64_t lddec,
                              float* aa_dec_out, float* R2, const int64_t* n_valid, void* stream);
/* dR2 (N,24,9) -> d_aa (N,72) += scale * J^T dR2[:,1:],  dR (N,24,9)[:,0] += scale * dR2[:,0]. */
int32_t nemo_v2v_prep_bwd(int64_t N, const float* aa, const float* dR2, float scale, float* d_aa,
                          float* dR, void* stream);

/* ------------------------------------------------------------------------------------------
 * Priors.
 * KL( N(mu, softplus(lv)) || N(0,1) ) summed over L, averaged over N
 * (vposer_model.py:48-56, nemo/neural_motion_model.py:2795-2802).  mulv (N, 2L) = [mu | lv].
 * scalar_out += kl;  d_mulv (N,2L) = upstream-free gradient (d kl / d [mu|lv]), overwritten. */
/* Loss read-back without a stream synchronisation (replaces the `.cpu()` read-backs of
 * neural_motion_model.py:3578-3584, which block the host until the optimiser step has finished): copies
 * src[0..n) (device, n <= 64) to host_dst and then stores 1 to *host_flag, both in PINNED host memory that
 * the device can address (hipHostMalloc; coherent), with system-scope release ordering.  The caller clears
 * *host_flag before enqueueing and polls it; kernels enqueued after this one keep running meanwhile. */
int32_t nemo_publish_scalars(const float* src, int32_t n, float* host_dst, int32_t* host_flag, void* stream);
int32_t nemo_kl_fwd_bwd(int64_t N, int64_t L, const float* mulv, int64_t ld, float* scalar_out,
                        float* d_mulv, int64_t ldd, const int64_t* n_valid, void* stream);
/* MaxMixturePrior (hmr/smplify/prior.py:181-196): per-sample min over M Gaussians, mean over N.
 * x (N, ldx) uses `dim` columns.  means (M,dim), precisions (M,dim,dim) SYMMETRIC (inverses of covariance
 * matrices; symmetrise (P+P^T)/2 on the host if in doubt), log_nllw (M) = log(nll_weights).
 * ws: scratch of N*M floats (per-component log-likelihoods).
 * scalar_out += mean;  per_sample (N) optional;  d_x (N, lddx) += scale * d mean / d x. */
int32_t nemo_gmm_fwd_bwd(int64_t N, int64_t M, int64_t dim, const float* x, int64_t ldx,
                         const float* means, const float* precisions, const float* log_nllw,
                         float* ws, float* scalar_out, float* per_sample, float scale, float* d_x,
                         int64_t lddx, const int64_t* n_valid, void* stream);
/* Robust 3-D pose loss used by warmup / NemoV3+ (:3489-3491, :3870-3882): mean over (N*dim) of
 * (mask>0.5) * GMoF(x - target).  scalar_out += mean; d_x += scale * grad. */
int32_t nemo_pose3d_fwd_bwd(int64_t N, int64_t dim, const float* x, int64_t ldx, const float* target,
                            const float* mask, const int64_t* view_idx, const int64_t* frame_idx,
                            int64_t T, float* scalar_out, float scale, float* d_x, int64_t lddx,
                            const int64_t* n_valid, void* stream);

/* ------------------------------------------------------------------------------------------
 * Fused multi-segment Adam / AdamW over one flat parameter buffer (torch.optim.Adam semantics,
 * betas (0.9,0.999), eps 1e-8; :3405-3431, :3588-3592).  Up to NEMO_ADAM_MAX_SEG segments per
 * launch; segment s covers [offset, offset+numel) of the flat buffers.
 */
#define NEMO_ADAM_MAX_SEG 16
typedef struct {
    int64_t offset, numel;
    float lr, weight_decay;
    float step_size;        /* lr / (1 - beta1^t), computed in double on the host */
    float bias_corr2_sqrt;  /* sqrt(1 - beta2^t) */
    int32_t adamw;                                        /* 0: coupled L2 (Adam), 1: decoupled */
    int32_t step;           /* t: updates applied to this segment so far (device tables: advanced by nemo_step_begin) */
} nemo_adam_seg;
int32_t nemo_adam_step(int32_t n_seg, const nemo_adam_seg* segs /* HOST */, float* params,
                       const float* grads, float* exp_avg, float* exp_avg_sq, float beta1, float beta2,
                       float eps, void* stream);
/* Same update with the segment table in DEVICE memory (n_seg entries; max_numel = largest segment):
 * the launch is then capturable in a HIP graph and replayed with per-step learning rates / bias
 * corrections refreshed by a small async copy. */
int32_t nemo_adam_step_dev(int32_t n_seg, const nemo_adam_seg* segs_dev, int64_t max_numel, float* params,
                           const float* grads, float* exp_avg, float* exp_avg_sq, float beta1, float beta2,
                           float eps, void* stream);

/* nemo_adam_step_dev that does NOTHING when *skip_if_nonzero != 0 (device scalar; NULL = unconditional): the update of
 * a captured warm-up step behind the device-side "nan gradient found" check of nemo/neural_motion_model.py:3497-3500 --
 * the reference stops BEFORE opt.step(), so the parameters must not receive the poisoned update. */
int32_t nemo_adam_step_dev_if(int32_t n_seg, const nemo_adam_seg* segs_dev, int64_t max_numel, float* params,
                              const float* grads, float* exp_avg, float* exp_avg_sq, float beta1, float beta2,
                              float eps, const float* skip_if_nonzero, void* stream);
/* *count_out += number of NaN entries of x[0..n): `torch.isnan(param.grad).any()` over every parameter (:3497-3500) as
 * one pass inside the captured step instead of a 9 MB scan plus a blocking read-back per warm-up iteration. */
int32_t nemo_nan_count(const float* x, int64_t n, float* count_out, void* stream);

/* Fit phases whose iterations the host does not have to see one by one (warmup :3455-3509, opt_cam :2869-2906: their
 * losses are only returned as a list at the end): the phase's batches are drawn up front in the reference's RNG order
 * and live in device memory as (steps, B) index tables, a device counter selects the row.
 * nemo_seq_gather: view_out[0..B) = all_view[*counter][0..B), frame_out likewise (the step's kernels then read the
 *                  usual index buffers: the captured graph of an iteration is the same for every iteration);
 * nemo_seq_log:    log[*counter][0..n) = src[0..n), then *counter += 1 (one block; the last launch of an iteration). */
int32_t nemo_seq_gather(const int64_t* all_view, const int64_t* all_frame, int64_t B, const int32_t* counter,
                        int64_t* view_out, int64_t* frame_out, void* stream);
int32_t nemo_seq_log(const float* src, int32_t n, float* log, int64_t ld, int32_t* counter, void* stream);

/* First launch of a step (replaces two memsets and the per-step host-to-device copy of the Adam table of the
 * graph-replayed step, `zero_grad()` x k + the bias corrections of :3586-3592): zero-fills the byte ranges
 * [z0, z0 + bytes0) and [z1, z1 + bytes1) (either may be NULL / 0; 16-byte aligned, multiples of 4 bytes) and, when
 * segs_dev != NULL, advances the n_seg entries of a DEVICE segment table by one update: step += 1,
 * step_size = lr / (1 - beta1^step), bias_corr2_sqrt = sqrt(1 - beta2^step) (float64 arithmetic, as torch computes
 * them on the host).  A later nemo_adam_step_dev of the same stream then applies update number `step`; the host only
 * re-uploads the table when a learning rate, the segment list or the step counts change under it. */
int32_t nemo_step_begin(void* z0, int64_t bytes0, void* z1, int64_t bytes1, nemo_adam_seg* segs_dev, int32_t n_seg,
                        double beta1, double beta2, void* stream);
/* nemo_phase_embed_fwd and nemo_step_begin (same arguments, same results) in ONE launch -- the first node of an
 * update step's graph: the zero-fills touch nothing the phase kernel reads or writes, so they run in further blocks of the
 * same grid and the step's critical chain is one dependent launch shorter. */
int32_t nemo_phase_embed_fwd_begin(int64_t N, int64_t V, int64_t T, int64_t K, int64_t D, int64_t C,
                                   const int64_t* view_idx, const int64_t* frame_idx, const float* raw_phase,
                                   const float* shifts, const float* scales, int64_t ldp, const float* log_sigmas,
                                   const float* codes, const float* code_noise, int32_t kernel_id, float* X, int64_t ldx,
                                   float* phase_out, float* den_out, float* x_meta, void* z0, int64_t bytes0, void* z1,
                                   int64_t bytes1, nemo_adam_seg* segs_dev, int32_t n_seg, double beta1, double beta2,
                                   int32_t n_absmax, const nemo_absmax_desc* absmax, void* stream);
/* (x_meta must lie OUTSIDE the two zero-filled ranges: their blocks run beside the phase blocks)
 * ABI 18: n_absmax (0 ... NEMO_CAST_XP_MAX) / absmax = a nemo_absmax_multi list that runs in the LEADING blocks of the same launch --
 * the pass over the split-precision chain's weights, which every update step needs in front of its cast launch (the weights changed):
 * beside the phase kernel on a second stream it cost a fork and a cross-queue join per step.  The records must lie outside the
 * zero-filled ranges too; nothing the other blocks write may be among the sources. */

/* Instance-code regulariser of NemoV3 / V4 (nemo/neural_motion_model.py:3864-3867):
 * scalar_out += mean(x[0..n)^2);  grad (may be NULL) += gscale * x. */
int32_t nemo_sqmean_fwd_bwd(int64_t n, const float* x, float* scalar_out, float* grad, float gscale, void* stream);

/* Small utilities. */
int32_t nemo_scale_neg_rowsum(int64_t N, int64_t cols, const float* X, int64_t ldx, float* out_row,
                              void* stream);  /* out_row[c] = -sum_s X[s][c]   (d trans_0) */

#ifdef __cplusplus
}
#endif
#endif /* NEMO_HIP_H */
