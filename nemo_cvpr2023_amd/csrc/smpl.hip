// SMPL kinematics, keypoint objective and full-mesh skinning kernels + the model context.
//
// Design (MI355X-first):
//  * the 2-D keypoint objective never touches the 6890-vertex mesh: every non-kinematic output
//    joint is a linear functional of the posed mesh, so nemo_ctx_create pre-contracts it with the
//    skinning weights and the pose blend-shapes into a (207 x nq*72) basis C1; per step one MFMA
//    GEMM (PF @ C1) + a 24-term sum per joint reproduces smplx's vertex pick / J_regressor_extra
//    exactly (up to fp32 re-association);
//  * the full mesh (VPoser v2v term, get_preds()['v']) is pose-blended by the MFMA GEMM and
//    skinned here with the 3x4 transforms fed from SGPRs (block-uniform) and the per-vertex
//    weights in VGPRs; the L1 loss and its gradient are produced in the same pass, the
//    vertex->joint gradient reduction runs on the matrix cores (16x16x4 f32 MFMA).
#include <vector>
#include <utility>
#include <cstring>
#include <cstdlib>
#include "common.h"
#include "rot_math.h"
#include "../../include/nemo_hip.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

#define NEMO_MAX_OUT 64

struct KpConst {
    int parents[24];
    int out_kind[NEMO_MAX_OUT];   // >=0: kinematic joint index; <0: -(q+1) mesh functional q
    int n_out, nq;
};

struct nemo_ctx {
    long NV;
    long NVp;                 // NV rounded up to a multiple of 16
    long ldP;                 // row stride of d_posedirs = 3*NVp; the array has 224 zero-padded rows so that
                              // the fused mesh kernel never needs a bounds predicate
    int n_out, nq;
    KpConst kc;
    // device constants
    float *d_posedirs, *d_v_shaped, *d_W, *d_Wt, *d_Jrest, *d_C1, *d_c0, *d_w0;
    // blend shapes rounded to bf16 (RNE) in MFMA-operand order for nemo_v2v_fused_bf16: [k-step S = 0..6][lane group
    // g = 0..3][vertex][component][8 consecutive k = 32 S + 8 g ..], zero for k >= 207 and for pad vertices
    unsigned short* d_posedirs_bf16;
    unsigned short* d_posedirs_sp3;    // three bf16 pieces per blend shape (mesh kernel MODE 4), [tile][S][component][piece][g][vertex][8 k]
    // skinning weights as TWO bf16 pieces (hi = bf16(w), lo = bf16(w - hi): 16 mantissa bits) for the split-precision
    // skinning of the bf16 mesh kernel, in MFMA-operand order:
    //   d_Wsk  [piece][vertex (NVp)][32]: forward A-operand rows (k = joint, zero for k >= 24)
    //   d_Wadj [tile][joint tile 2][piece][lane 64][8]: adjoint A-operand of v_mfma_f32_16x16x32_bf16 -- lane (l15, g)
    //          holds W[16 tile + 4 g + t][16 jt + l15] for t = 0..3 TWICE (k = 8 g + t and 8 g + 4 + t: the B-operand
    //          carries the hi pieces of dT in the first four k and the lo pieces in the last four)
    unsigned short *d_Wsk, *d_Wadj;
    unsigned short* d_posedirs_sph;    // two fp16 pieces per blend shape x sph_scale (mesh kernel MODE 5), [tile][S][component][piece][lane][8 k]
    unsigned short* d_Wadjh;           // MODE 5: [tile][joint tile 2][variant 2: W0|W0, W1|0][lane 64][8] (two fp16 pieces of 2^14 W)
    float sph_scale;                   // power of two: max |P| * sph_scale in [2^13, 2^14)
    // range guard of MODE 5 (two fp16 pieces): the kernel stages the blended vertices vp = v_shaped + P pf and their pieces x 2^12,
    // |pf| <= 2 (entries of R - I) -- vp_bound = max_v (|v_shaped| + 2 sum_k |P[k][v]|) bounds every |vp|; MODE 5 is used only while
    // vp_bound * 2^12 < 2^15.9 (else MODE 4: three bf16 pieces, fp32's exponent range).  h_pabs2 = 2 sum_k |P[k][.]| per coordinate.
    std::vector<float> h_pabs2;
    float vp_bound = 0.f;
    int split_ok = 1;
    // MODE 6: the relative transforms' entries are staged as fp16 pieces of 2^12 A: |A| <= 1 for the rotation part, <= |J_0| + the longest
    // chain of bone lengths + max |J| for the translation part (a_bound, from the rest joints: recomputed by nemo_ctx_set_betas)
    float a_bound = 0.f;
    int skin_mfma_ok = 1;
    unsigned short* d_Wskh = nullptr;   // three fp16 pieces of 2^14 W, [piece][NVp][32 joints (24 + zero tail)]
    unsigned short* d_Wadj3;           // MODE 4: [tile][joint tile 2][variant 3: W0|W0, W1|W1, W0|W2][lane 64][8] (three bf16 pieces of W)
    // SPARSE skinning weights (the published SMPL model has at most four non-zero weights per vertex; the dense 24-column
    // product of lbs.py:236-241 then multiplies 20 zeros per vertex): per vertex (NVp of them, zero rows for the pad) the
    // <= 4 non-zero weights in ascending joint order, and their joints as four bytes holding 3 * joint (the offset of the
    // joint's 3 x 4 transform in float4 units).  skin_nnz = the largest number of non-zero weights any vertex has;
    // the mesh kernel takes its sparse-skinning form when skin_nnz <= 4 and skin_sparse is set (the default then).
    float* d_Wsp_w;
    unsigned int* d_Wsp_j;
    int skin_nnz, skin_sparse;
    // host copies needed to re-derive the shape-dependent constants
    std::vector<float> h_v_template, h_shapedirs, h_Jreg, h_W;
    std::vector<std::vector<std::pair<long, float>>> q_rows;   // sparse rows of the nq functionals
};

// ------------------------------------------------------------------------------------------ ctx
#define HIPCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) return (int32_t)e_; } while (0)

extern "C" int32_t nemo_ctx_set_betas(nemo_ctx* ctx, const float* betas) {
    if (!ctx) return NEMO_EINVAL;
    const long NV = ctx->NV;
    float b[10] = {0};
    if (betas) memcpy(b, betas, sizeof(b));
    std::vector<float> vs(NV * 3);
    for (long v = 0; v < NV; ++v)
        for (int c = 0; c < 3; ++c) {
            double acc = 0.0;   // lbs.py:299  einsum('bl,mkl->bmk')
            for (int l = 0; l < 10; ++l) acc += (double)b[l] * ctx->h_shapedirs[(v * 3 + c) * 10 + l];
            vs[v * 3 + c] = ctx->h_v_template[v * 3 + c] + (float)acc;
        }
    {
        float bnd = 0.f;
        for (long i = 0; i < NV * 3; ++i) bnd = fmaxf(bnd, fabsf(vs[i]) + (ctx->h_pabs2.empty() ? 0.f : ctx->h_pabs2[i]));
        ctx->vp_bound = bnd;
        ctx->split_ok = (bnd * 4096.f < 61000.f) ? 1 : 0;      // 2^15.9 = 61 147; not finite: 0
    }
    std::vector<float> J(72);
    for (int j = 0; j < 24; ++j)
        for (int c = 0; c < 3; ++c) {
            double acc = 0.0;   // lbs.py:274  einsum('bik,ji->bjk')
            for (long v = 0; v < NV; ++v) acc += (double)ctx->h_Jreg[j * NV + v] * vs[v * 3 + c];
            J[j * 3 + c] = (float)acc;
        }
    {
        float path[24], mj = 0.f, worst = 0.f;
        auto nrm = [&](int a) { return sqrtf(J[a * 3] * J[a * 3] + J[a * 3 + 1] * J[a * 3 + 1] + J[a * 3 + 2] * J[a * 3 + 2]); };
        for (int j = 0; j < 24; ++j) {
            mj = fmaxf(mj, nrm(j));
            if (j == 0) { path[0] = nrm(0); continue; }
            const int p = ctx->kc.parents[j];
            const float dx = J[j * 3] - J[p * 3], dy = J[j * 3 + 1] - J[p * 3 + 1], dz = J[j * 3 + 2] - J[p * 3 + 2];
            path[j] = path[p] + sqrtf(dx * dx + dy * dy + dz * dz);
            worst = fmaxf(worst, path[j]);
        }
        ctx->a_bound = fmaxf(1.f, worst + mj);
        ctx->skin_mfma_ok = (ctx->split_ok && ctx->a_bound * 4096.f < 61000.f) ? 1 : 0;
    }
    std::vector<float> c0((size_t)ctx->nq * 72 + 1, 0.f);
    for (int q = 0; q < ctx->nq; ++q)
        for (int j = 0; j < 24; ++j) {
            double acc[3] = {0, 0, 0};
            for (auto& e : ctx->q_rows[q]) {
                const double w = (double)e.second * ctx->h_W[e.first * 24 + j];
                for (int c = 0; c < 3; ++c) acc[c] += w * vs[e.first * 3 + c];
            }
            for (int c = 0; c < 3; ++c) c0[q * 72 + j * 3 + c] = (float)acc[c];
        }
    HIPCHK(hipMemcpy(ctx->d_v_shaped, vs.data(), sizeof(float) * NV * 3, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(ctx->d_Jrest, J.data(), sizeof(float) * 72, hipMemcpyHostToDevice));
    if (ctx->nq)
        HIPCHK(hipMemcpy(ctx->d_c0, c0.data(), sizeof(float) * ctx->nq * 72, hipMemcpyHostToDevice));
    return NEMO_OK;
}

extern "C" int32_t nemo_ctx_create(nemo_ctx** out, int64_t NV, const float* v_template,
                                   const float* shapedirs, const float* posedirs, const float* J_regressor,
                                   const float* lbs_weights, const int64_t* parents, int64_t n_extra,
                                   const float* J_regressor_extra, int64_t n_sel,
                                   const int64_t* sel_vertex_ids, int64_t n_out, const int64_t* out_joints) {
    if (!out || NV <= 0 || !v_template || !shapedirs || !posedirs || !J_regressor || !lbs_weights ||
        !parents || n_out < 0 || n_out > NEMO_MAX_OUT || (n_out && !out_joints))
        return NEMO_EINVAL;
    if (parents[0] >= 0) return NEMO_EINVAL;
    for (int i = 1; i < 24; ++i)
        if (parents[i] < 0 || parents[i] >= i) return NEMO_EINVAL;   // topological order required
    nemo_ctx* c = new nemo_ctx();
    c->NV = NV;
    c->NVp = ((NV + 15) / 16) * 16;          // vertices padded to whole 16-vertex MFMA tiles
    c->ldP = c->NVp * 3;                     // multiple of 48: 16-byte aligned rows
    c->n_out = (int)n_out;
    for (int i = 0; i < 24; ++i) c->kc.parents[i] = (int)parents[i];
    c->h_v_template.assign(v_template, v_template + NV * 3);
    c->h_shapedirs.assign(shapedirs, shapedirs + NV * 30);
    c->h_pabs2.assign((size_t)NV * 3, 0.f);
    for (int k = 0; k < 207; ++k)
        for (long i = 0; i < NV * 3; ++i) c->h_pabs2[i] += 2.f * fabsf(posedirs[(size_t)k * NV * 3 + i]);
    c->h_Jreg.assign(J_regressor, J_regressor + 24 * NV);
    c->h_W.assign(lbs_weights, lbs_weights + NV * 24);
    // classify the output joints
    int nq = 0;
    for (int o = 0; o < n_out; ++o) {
        const long idx = out_joints[o];
        if (idx < 0 || idx >= 24 + n_sel + n_extra) { delete c; return NEMO_EINVAL; }
        if (idx < 24) { c->kc.out_kind[o] = (int)idx; continue; }
        std::vector<std::pair<long, float>> row;
        if (idx < 24 + n_sel) {
            const long v = sel_vertex_ids[idx - 24];
            if (v < 0 || v >= NV) { delete c; return NEMO_EINVAL; }
            row.push_back({v, 1.0f});                               // smplx VertexJointSelector pick
        } else {
            const float* r = J_regressor_extra + (idx - 24 - n_sel) * NV;   // hmr/smpl.py:33-34
            for (long v = 0; v < NV; ++v)
                if (r[v] != 0.f) row.push_back({v, r[v]});
        }
        c->q_rows.push_back(std::move(row));
        c->kc.out_kind[o] = -(nq + 1);
        ++nq;
    }
    c->nq = nq;
    c->kc.n_out = (int)n_out;
    c->kc.nq = nq;

    // pre-contraction: w0[q][j] = sum_v r_q[v] W[v][j];  C1[p][q][j][c] = sum_v r_q[v] W[v][j] P[p][3v+c]
    std::vector<float> w0((size_t)nq * 24 + 1, 0.f), C1((size_t)207 * nq * 72 + 1, 0.f);
    for (int q = 0; q < nq; ++q) {
        const auto& row = c->q_rows[q];
        for (int j = 0; j < 24; ++j) {
            double acc = 0.0;
            for (auto& e : row) acc += (double)e.second * lbs_weights[e.first * 24 + j];
            w0[q * 24 + j] = (float)acc;
        }
        std::vector<double> acc(72);
        for (int p = 0; p < 207; ++p) {
            std::fill(acc.begin(), acc.end(), 0.0);
            const float* Pp = posedirs + (size_t)p * NV * 3;
            for (auto& e : row) {
                const float* pv = Pp + e.first * 3;
                const float* wv = lbs_weights + e.first * 24;
                for (int j = 0; j < 24; ++j) {
                    const double w = (double)e.second * wv[j];
                    acc[j * 3 + 0] += w * pv[0];
                    acc[j * 3 + 1] += w * pv[1];
                    acc[j * 3 + 2] += w * pv[2];
                }
            }
            for (int k = 0; k < 72; ++k) C1[((size_t)p * nq + q) * 72 + k] = (float)acc[k];
        }
    }
    std::vector<float> Wt((size_t)24 * NV);
    for (long v = 0; v < NV; ++v)
        for (int j = 0; j < 24; ++j) Wt[j * NV + v] = lbs_weights[v * 24 + j];

    c->d_posedirs = c->d_v_shaped = c->d_W = c->d_Wt = c->d_Jrest = c->d_C1 = c->d_c0 = c->d_w0 = nullptr;
#define ALLOC(p, n) HIPCHK(hipMalloc((void**)&(p), sizeof(float) * (size_t)((n) > 0 ? (n) : 1)))
    ALLOC(c->d_posedirs, 224 * c->ldP); ALLOC(c->d_v_shaped, c->NVp * 3); ALLOC(c->d_W, c->NVp * 24);
    ALLOC(c->d_Wt, NV * 24); ALLOC(c->d_Jrest, 72); ALLOC(c->d_C1, 207 * nq * 72);
    ALLOC(c->d_c0, nq * 72); ALLOC(c->d_w0, nq * 24);
#undef ALLOC
    HIPCHK(hipMemset(c->d_posedirs, 0, sizeof(float) * 224 * c->ldP));
    HIPCHK(hipMemset(c->d_v_shaped, 0, sizeof(float) * c->NVp * 3));
    HIPCHK(hipMemset(c->d_W, 0, sizeof(float) * c->NVp * 24));
    HIPCHK(hipMemcpy2D(c->d_posedirs, sizeof(float) * c->ldP, posedirs, sizeof(float) * NV * 3,
                       sizeof(float) * NV * 3, 207, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(c->d_W, lbs_weights, sizeof(float) * NV * 24, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(c->d_Wt, Wt.data(), sizeof(float) * NV * 24, hipMemcpyHostToDevice));
    {
        auto bf16 = [](float f) -> unsigned short {
            unsigned int u; memcpy(&u, &f, 4);
            if ((u & 0x7fffffffu) > 0x7f800000u) return (unsigned short)((u >> 16) | 0x40);      // NaN
            return (unsigned short)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);                    // round to nearest even
        };
        // [vertex tile][k-step S of 32][component][g = 8-k group][vertex in tile][8 k]: the 64 lanes of ONE buffer_load_dwordx4 of
        // the mesh kernel (lane = 16 g + vertex) read 1 KB of consecutive memory, a tile's 21 loads 21 KB.  (Until round 5:
        // [S][g][vertex][component][8 k] -- a load took 16 bytes of every 48: 26 cache lines touched per instruction, the L1's
        // tag / data-return path 66 - 79 % busy over the whole kernel, profiles/r05_pmc_mesh_b16.md.)
        std::vector<unsigned short> pb((size_t)28 * c->NVp * 24, 0);
        for (int k = 0; k < 207; ++k) {
            const int S = k / 32, g = (k % 32) / 8, i = k % 8;
            const float* Pk = posedirs + (size_t)k * NV * 3;
            for (long v = 0; v < NV; ++v)
                for (int cc = 0; cc < 3; ++cc)
                    pb[(((((size_t)(v / 16) * 7 + S) * 3 + cc) * 4 + g) * 16 + v % 16) * 8 + i] = bf16(Pk[v * 3 + cc]);
        }
        c->d_posedirs_bf16 = nullptr;
        HIPCHK(hipMalloc((void**)&c->d_posedirs_bf16, pb.size() * 2));
        HIPCHK(hipMemcpy(c->d_posedirs_bf16, pb.data(), pb.size() * 2, hipMemcpyHostToDevice));
        {
            // fp32-equivalent split: x = x0 + x1 + x2 with x0 = bf16(x), x1 = bf16(x - x0), x2 = bf16(x - x0 - x1) (the
            // subtractions are exact in fp32)
            auto unb = [](unsigned short h) -> float { unsigned int u = (unsigned int)h << 16; float f; memcpy(&f, &u, 4); return f; };
            std::vector<unsigned short> p3((size_t)28 * c->NVp * 72, 0);
            for (int k = 0; k < 207; ++k) {
                const int S = k / 32, g = (k % 32) / 8, i = k % 8;
                const float* Pk = posedirs + (size_t)k * NV * 3;
                for (long v = 0; v < NV; ++v)
                    for (int cc = 0; cc < 3; ++cc) {
                        float x = Pk[v * 3 + cc];
                        for (int pc = 0; pc < 3; ++pc) {
                            const unsigned short h = bf16(x);
                            p3[((((((size_t)(v / 16) * 7 + S) * 3 + cc) * 3 + pc) * 4 + g) * 16 + v % 16) * 8 + i] = h;
                            x -= unb(h);
                        }
                    }
            }
            c->d_posedirs_sp3 = nullptr;
            HIPCHK(hipMalloc((void**)&c->d_posedirs_sp3, p3.size() * 2));
            HIPCHK(hipMemcpy(c->d_posedirs_sp3, p3.data(), p3.size() * 2, hipMemcpyHostToDevice));
        }
        {
            // two fp16 pieces of s P: x0 = fp16(s x), x1 = fp16(s x - x0), s the power of two that puts max |P| into [2^13, 2^14):
            // x0 + x1 = s x up to 2^-23 |s x| wherever x1 is a normal fp16 (|x| >= 2^-16 max |P|; below that the absolute error is
            // <= 2^-24, i.e. 2^-38 of the largest blend shape)
            float pmax = 0.f;
            for (size_t i = 0; i < (size_t)207 * NV * 3; ++i) pmax = fmaxf(pmax, fabsf(posedirs[i]));
            int ex = 0;
            if (pmax > 0.f) (void)frexpf(pmax, &ex);            // pmax = m 2^ex, m in [0.5, 1)
            c->sph_scale = ldexpf(1.f, 14 - ex);                // max |P| * scale in [2^13, 2^14)
            auto f16b = [](float f) -> unsigned short { const _Float16 h = (_Float16)f; unsigned short b; memcpy(&b, &h, 2); return b; };
            auto unf16 = [](unsigned short b) -> float { _Float16 h; memcpy(&h, &b, 2); return (float)h; };
            std::vector<unsigned short> p2((size_t)28 * c->NVp * 48, 0);
            for (int k = 0; k < 207; ++k) {
                const int S = k / 32, g = (k % 32) / 8, i = k % 8;
                const float* Pk = posedirs + (size_t)k * NV * 3;
                for (long v = 0; v < NV; ++v)
                    for (int cc = 0; cc < 3; ++cc) {
                        float x = Pk[v * 3 + cc] * c->sph_scale;
                        for (int pc = 0; pc < 2; ++pc) {
                            const unsigned short h = f16b(x);
                            p2[((((((size_t)(v / 16) * 7 + S) * 3 + cc) * 2 + pc) * 4 + g) * 16 + v % 16) * 8 + i] = h;
                            x -= unf16(h);
                        }
                    }
            }
            c->d_posedirs_sph = nullptr;
            HIPCHK(hipMalloc((void**)&c->d_posedirs_sph, p2.size() * 2));
            HIPCHK(hipMemcpy(c->d_posedirs_sph, p2.data(), p2.size() * 2, hipMemcpyHostToDevice));
        }
        auto unbf = [](unsigned short h) -> float { unsigned int u = (unsigned int)h << 16; float f; memcpy(&f, &u, 4); return f; };
        const long ntl = c->NVp / 16;
        std::vector<unsigned short> wsk((size_t)2 * c->NVp * 32, 0), wadj((size_t)ntl * 2 * 2 * 64 * 8, 0);
        std::vector<unsigned short> wadj3((size_t)ntl * 2 * 3 * 64 * 8, 0), wadjh((size_t)ntl * 2 * 2 * 64 * 8, 0);
        auto f16w = [](float f) -> unsigned short { const _Float16 h = (_Float16)f; unsigned short b; memcpy(&b, &h, 2); return b; };
        auto unf16w = [](unsigned short b) -> float { _Float16 h; memcpy(&h, &b, 2); return (float)h; };
        for (long v = 0; v < NV; ++v)
            for (int j = 0; j < 24; ++j) {
                const float w = lbs_weights[v * 24 + j];
                const unsigned short hi = bf16(w), lo = bf16(w - unbf(hi));
                wsk[((size_t)0 * c->NVp + v) * 32 + j] = hi;
                wsk[((size_t)1 * c->NVp + v) * 32 + j] = lo;
                const long t = v / 16; const int vv = (int)(v % 16), g = vv / 4, r = vv % 4, jt = j / 16, l15 = j % 16;
                const size_t base = ((((size_t)t * 2 + jt) * 2) * 64 + (g * 16 + l15)) * 8;
                wadj[base + r] = hi; wadj[base + 4 + r] = hi;
                wadj[base + 64 * 8 + r] = lo; wadj[base + 64 * 8 + 4 + r] = lo;
                // three pieces (fp32-equivalent adjoint of MODE 4): A images [W0 | W0], [W1 | W1], [W0 | W2] over the two
                // 16-vertex piece slots of a K = 32 instruction
                const unsigned short w0 = hi, w1 = lo, w2 = bf16(w - unbf(hi) - unbf(lo));
                const size_t b3 = ((((size_t)t * 2 + jt) * 3) * 64 + (g * 16 + l15)) * 8;
                wadj3[b3 + r] = w0; wadj3[b3 + 4 + r] = w0;
                wadj3[b3 + 64 * 8 + r] = w1; wadj3[b3 + 64 * 8 + 4 + r] = w1;
                wadj3[b3 + 2 * 64 * 8 + r] = w0; wadj3[b3 + 2 * 64 * 8 + 4 + r] = w2;
                // two fp16 pieces of 2^14 W (MODE 5): A images [W0 | W0], [W1 | 0]
                const unsigned short h0 = f16w(w * 16384.f), h1 = f16w(w * 16384.f - unf16w(f16w(w * 16384.f)));
                const size_t bh = ((((size_t)t * 2 + jt) * 2) * 64 + (g * 16 + l15)) * 8;
                wadjh[bh + r] = h0; wadjh[bh + 4 + r] = h0;
                wadjh[bh + 64 * 8 + r] = h1;
            }
        {
            // THREE fp16 pieces of 2^14 W (33 bits: the weights exactly) -- they live in registers; the transforms keep two (LDS)
            std::vector<unsigned short> wskh((size_t)3 * c->NVp * 32, 0);
            for (long v = 0; v < NV; ++v)
                for (int j = 0; j < 24; ++j) {
                    float x = lbs_weights[v * 24 + j] * 16384.f;
                    for (int pc = 0; pc < 3; ++pc) {
                        const unsigned short h = f16w(x);
                        wskh[((size_t)pc * c->NVp + v) * 32 + j] = h;
                        x -= unf16w(h);
                    }
                }
            HIPCHK(hipMalloc((void**)&c->d_Wskh, wskh.size() * 2));
            HIPCHK(hipMemcpy(c->d_Wskh, wskh.data(), wskh.size() * 2, hipMemcpyHostToDevice));
        }
        c->d_Wsk = c->d_Wadj = nullptr;
        HIPCHK(hipMalloc((void**)&c->d_Wsk, wsk.size() * 2));
        HIPCHK(hipMemcpy(c->d_Wsk, wsk.data(), wsk.size() * 2, hipMemcpyHostToDevice));
        HIPCHK(hipMalloc((void**)&c->d_Wadj, wadj.size() * 2));
        HIPCHK(hipMemcpy(c->d_Wadj, wadj.data(), wadj.size() * 2, hipMemcpyHostToDevice));
        c->d_Wadj3 = nullptr;
        HIPCHK(hipMalloc((void**)&c->d_Wadj3, wadj3.size() * 2));
        HIPCHK(hipMemcpy(c->d_Wadj3, wadj3.data(), wadj3.size() * 2, hipMemcpyHostToDevice));
        c->d_Wadjh = nullptr;
        HIPCHK(hipMalloc((void**)&c->d_Wadjh, wadjh.size() * 2));
        HIPCHK(hipMemcpy(c->d_Wadjh, wadjh.data(), wadjh.size() * 2, hipMemcpyHostToDevice));
    }
    {
        int nnz_max = 0;
        std::vector<float> sw((size_t)c->NVp * 4, 0.f);
        std::vector<unsigned int> sj((size_t)c->NVp, 0u);
        for (long v = 0; v < NV; ++v) {
            int n = 0;
            for (int j = 0; j < 24; ++j) {
                const float w = lbs_weights[v * 24 + j];
                if (w == 0.f) continue;
                if (n < 4) { sw[v * 4 + n] = w; sj[v] |= (unsigned)(3 * j) << (8 * n); }
                ++n;
            }
            if (n > nnz_max) nnz_max = n;
        }
        c->skin_nnz = nnz_max;
        c->skin_sparse = nnz_max <= 4;
        c->d_Wsp_w = nullptr; c->d_Wsp_j = nullptr;
        HIPCHK(hipMalloc((void**)&c->d_Wsp_w, sw.size() * 4));
        HIPCHK(hipMemcpy(c->d_Wsp_w, sw.data(), sw.size() * 4, hipMemcpyHostToDevice));
        HIPCHK(hipMalloc((void**)&c->d_Wsp_j, sj.size() * 4));
        HIPCHK(hipMemcpy(c->d_Wsp_j, sj.data(), sj.size() * 4, hipMemcpyHostToDevice));
    }
    if (nq) {
        HIPCHK(hipMemcpy(c->d_C1, C1.data(), sizeof(float) * 207 * nq * 72, hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(c->d_w0, w0.data(), sizeof(float) * nq * 24, hipMemcpyHostToDevice));
    }
    const int32_t rc = nemo_ctx_set_betas(c, nullptr);
    if (rc) return rc;
    *out = c;
    return NEMO_OK;
}

extern "C" int32_t nemo_ctx_destroy(nemo_ctx* c) {
    if (!c) return NEMO_OK;
    float* ptrs[] = {c->d_posedirs, c->d_v_shaped, c->d_W, c->d_Wt, c->d_Jrest, c->d_C1, c->d_c0, c->d_w0};
    for (float* p : ptrs)
        if (p) (void)hipFree(p);
    if (c->d_posedirs_bf16) (void)hipFree(c->d_posedirs_bf16);
    if (c->d_posedirs_sp3) (void)hipFree(c->d_posedirs_sp3);
    if (c->d_Wsk) (void)hipFree(c->d_Wsk);
    if (c->d_Wadj) (void)hipFree(c->d_Wadj);
    if (c->d_Wadj3) (void)hipFree(c->d_Wadj3);
    if (c->d_Wadjh) (void)hipFree(c->d_Wadjh);
    if (c->d_Wskh) (void)hipFree(c->d_Wskh);
    if (c->d_posedirs_sph) (void)hipFree(c->d_posedirs_sph);
    if (c->d_Wsp_w) (void)hipFree(c->d_Wsp_w);
    if (c->d_Wsp_j) (void)hipFree(c->d_Wsp_j);
    delete c;
    return NEMO_OK;
}
extern "C" int64_t nemo_ctx_num_verts(const nemo_ctx* c) { return c ? c->NV : -1; }
extern "C" int32_t nemo_ctx_skin_nnz(const nemo_ctx* c) { return c ? c->skin_nnz : -1; }
extern "C" int32_t nemo_ctx_split_ok(const nemo_ctx* c) { return c ? c->split_ok : -1; }
extern "C" int32_t nemo_ctx_skin_mfma_ok(const nemo_ctx* c) { return c ? c->skin_mfma_ok : -1; }
extern "C" float nemo_ctx_vp_bound(const nemo_ctx* c) { return c ? c->vp_bound : -1.f; }
extern "C" int32_t nemo_ctx_skin_sparse(const nemo_ctx* c) { return c ? c->skin_sparse : -1; }
extern "C" int32_t nemo_ctx_set_skin_sparse(nemo_ctx* c, int32_t enable) {
    if (!c || (enable && c->skin_nnz > 4)) return NEMO_EINVAL;
    c->skin_sparse = enable ? 1 : 0;
    return NEMO_OK;
}
extern "C" int64_t nemo_ctx_nq(const nemo_ctx* c) { return c ? c->nq : -1; }
extern "C" const float* nemo_ctx_C1(const nemo_ctx* c) { return c ? c->d_C1 : nullptr; }
extern "C" const float* nemo_ctx_c0(const nemo_ctx* c) { return c ? c->d_c0 : nullptr; }
extern "C" const float* nemo_ctx_posedirs(const nemo_ctx* c) { return c ? c->d_posedirs : nullptr; }
extern "C" int64_t nemo_ctx_posedirs_ld(const nemo_ctx* c) { return c ? c->ldP : -1; }
extern "C" const float* nemo_ctx_v_shaped(const nemo_ctx* c) { return c ? c->d_v_shaped : nullptr; }

typedef unsigned int u32x3 __attribute__((ext_vector_type(3)));

namespace {

// ------------------------------------------------------------------------------------------ FK
// 64 bodies per block, 256 threads.  The kinematic chain is inherently sequential per body (23
// dependent 3x3 products), so one thread per body walks it -- but at LDS latency: phase 1 stages the
// block's rotations (FK_TB x 216 contiguous floats) with coalesced loads into LDS rows of stride 217
// (conflict-free per-body reads), phase 2 runs the chains with their state in LDS as
// [joint*12 + e][lane], phase 3 writes transforms / posed joints back with coalesced stores.  A thread-
// per-body kernel working directly on global memory issues ~50 scattered accesses per joint (64 cache
// lines each) and was 4x slower.
#ifndef FK_TB
#define FK_TB 16      // bodies per block (256 threads; 32 KB of LDS): measured 64 -> 32 -> 16 bodies:
                      // fwd 18.7 -> 14.6 -> 12.6 us, bwd 29.9 -> 24.0 -> 22.0 us (more, smaller blocks: all CUs busy)
#endif
#define FK_GS (FK_TB + 1)       // chain-state stride: [entry][lane] with +1 pad -> conflict-free for fixed entry (phase 2)
                       // AND for consecutive entries of one body (coalesced phases 1/3)
#define FK_RS 217
#define FK_IT(n) (((n) + 255) / 256)        // compile-time trip count of a 256-thread copy loop over n items
#define FK_LDS_BYTES ((24 * 12 * FK_GS + FK_TB * FK_RS) * (int)sizeof(float))

// Copy loops have COMPILE-TIME trip counts and issue all their (unconditional, clamped) loads before
// the first use: a rolled runtime-bound loop waits one full memory round trip per iteration, which
// made the copy phases 5x longer than the kinematic chain itself.
__device__ __forceinline__ void fk_load_R(const float* __restrict__ R, long row0, int nb, float* Rl) {
    // FK_TB bodies x 216 floats = FK_TB * 54 float4
    const float4* src = reinterpret_cast<const float4*>(R + row0 * 216);
    const int lim = nb * 54;
    float4 v[FK_IT(FK_TB * 54)];
#pragma unroll
    for (int it = 0; it < FK_IT(FK_TB * 54); ++it) {
        const int i4 = it * 256 + threadIdx.x;
        v[it] = src[i4 < lim ? i4 : 0];
    }
#pragma unroll
    for (int it = 0; it < FK_IT(FK_TB * 54); ++it) {
        const int i4 = it * 256 + threadIdx.x;
        if (i4 < lim) {
            const int bdy = i4 / 54, k = (i4 % 54) * 4;
            float* d = Rl + bdy * FK_RS + k;
            d[0] = v[it].x; d[1] = v[it].y; d[2] = v[it].z; d[3] = v[it].w;
        }
    }
}

__global__ __launch_bounds__(256) void fk_fwd_kernel(long rows, const float* __restrict__ R,
                                                     const float* __restrict__ Jrest, KpConst kc,
                                                     float* __restrict__ A, float* __restrict__ Jp,
                                                     float* __restrict__ PF, long ldpf) {
    extern __shared__ float lds[];
    float* G = lds;                                  // [(j*12 + e) * FK_GS + lane]: e<9 rotation, e>=9 translation
    float* Rl = lds + 24 * 12 * FK_GS;               // [body][FK_RS]
    __shared__ float Js[72];                         // rest joints and parents: the chain must not wait on
    __shared__ int Ps[24];                           // global / kernarg loads between dependent steps
    const int tid = threadIdx.x;
    const long row0 = (long)blockIdx.x * FK_TB;
    const int nb = (int)min((long)FK_TB, rows - row0);
    if (tid < 72) Js[tid] = Jrest[tid];
    if (tid < 24) Ps[tid] = kc.parents[tid];
    // ---- phase 1: coalesced load of the rotations
    fk_load_R(R, row0, nb, Rl);
    __syncthreads();
    // pose feature (R[1:] - I), written from LDS with coalesced stores: 64 x 207 floats
    if (PF) {
#pragma unroll
        for (int it = 0; it < FK_IT(FK_TB * 207); ++it) {
            const int idx = it * 256 + tid;
            const int bdy = idx / 207, k = idx % 207;
            if (bdy < nb) PF[(row0 + bdy) * ldpf + k] = Rl[bdy * FK_RS + 9 + k] - ((k % 9) % 4 == 0 ? 1.f : 0.f);
        }
    }
    // ---- phase 2: the chain, 16 lanes per body (12 active: one per element of [G_R | G_t]); a body's lanes sit in one
    // wave, so a joint's elements are visible to the next joint's reads by LDS program order alone -- no block barrier
    // inside the chain.  (One thread per body spent ~550 cycles per joint on 21 dependent LDS reads and 12 writes:
    // 5.6 of the kernel's 10.4 us; tools/bench_fk_graph.py.)
    static_assert(FK_TB * 16 == 256, "16 lanes per body");
    {
        const int bdy = tid >> 4, e = tid & 15;
        if (bdy < nb && e < 12) {
            const float* Rr = Rl + bdy * FK_RS;
            const int r = e < 9 ? e / 3 : e - 9, c = e % 3;
#define GL(j, k) G[((j) * 12 + (k)) * FK_GS + bdy]
            GL(0, e) = e < 9 ? Rr[e] : Js[e - 9];
            // (fully unrolled: the parent ids, rest offsets and the lane's column of every R_i are loaded ahead of the
            //  chain -- only the parent's row read, three FMAs and the write remain dependent)
#pragma unroll
            for (int i = 1; i < 24; ++i) {
                const int p = Ps[i];
                const float g0 = GL(p, r * 3), g1 = GL(p, r * 3 + 1), g2 = GL(p, r * 3 + 2);
                float val;
                if (e < 9) val = g0 * Rr[i * 9 + c] + g1 * Rr[i * 9 + 3 + c] + g2 * Rr[i * 9 + 6 + c];
                else val = g0 * (Js[i * 3] - Js[p * 3]) + g1 * (Js[i * 3 + 1] - Js[p * 3 + 1]) +
                           g2 * (Js[i * 3 + 2] - Js[p * 3 + 2]) + GL(p, 9 + r);
                GL(i, e) = val;
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_wave_barrier();
            }
#undef GL
        }
    }
    __syncthreads();
    // ---- phase 3: coalesced float4 write-out.  A[j] row r = [G_R[r] | G_t[r] - G_R[r].J_j] (lbs.py:399-402)
    float4* A4 = reinterpret_cast<float4*>(A + row0 * 288);
#pragma unroll
    for (int it = 0; it < FK_IT(FK_TB * 72); ++it) {   // FK_TB x 72 float4
        const int i4 = it * 256 + tid;
        const int bdy = i4 / 72, jr = i4 % 72, j = jr / 3, r = jr % 3;
        const float* g = G + (j * 12) * FK_GS + bdy;
        const float g0 = g[(r * 3) * FK_GS], g1 = g[(r * 3 + 1) * FK_GS], g2 = g[(r * 3 + 2) * FK_GS];
        const float t = g[(9 + r) * FK_GS] - (g0 * Js[j * 3] + g1 * Js[j * 3 + 1] + g2 * Js[j * 3 + 2]);
        if (bdy < nb) A4[i4] = make_float4(g0, g1, g2, t);
    }
    float4* J4 = reinterpret_cast<float4*>(Jp + row0 * 72);
#pragma unroll
    for (int it = 0; it < FK_IT(FK_TB * 18); ++it) {   // FK_TB x 18 float4
        const int i4 = it * 256 + tid;
        const int bdy = i4 / 18, e0 = (i4 % 18) * 4;
        if (i4 < FK_TB * 18 && bdy < nb) {
            float v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] = G[(((e0 + u) / 3) * 12 + 9 + (e0 + u) % 3) * FK_GS + bdy];
            J4[i4] = make_float4(v[0], v[1], v[2], v[3]);
        }
    }
}

__global__ __launch_bounds__(256) void fk_bwd_kernel(long rows, const float* __restrict__ R,
                                                     const float* __restrict__ A,
                                                     const float* __restrict__ Jrest, KpConst kc,
                                                     const float* __restrict__ dA,
                                                     const float* __restrict__ dJp,
                                                     const float* __restrict__ dPF, long lddpf,
                                                     float* __restrict__ dR) {
    extern __shared__ float lds[];
    float* D = lds;                                  // dG accumulators [(j*12 + r*4 + c) * FK_GS + lane]
    float* Rl = lds + 24 * 12 * FK_GS;               // rotations in, rotation gradients out (in place)
    __shared__ float Js[72];
    __shared__ int Ps[24];
    const int tid = threadIdx.x;
    const long row0 = (long)blockIdx.x * FK_TB;
    const int nb = (int)min((long)FK_TB, rows - row0);
    if (tid < 72) Js[tid] = Jrest[tid];
    if (tid < 24) Ps[tid] = kc.parents[tid];
    __syncthreads();
    // ---- phase 1: coalesced loads.  dA -> dG:  A_t = G_t - G_R J  =>  dG_R = dA_R - dA_t (x) J ; dG_t = dA_t (+ dJp)
    fk_load_R(R, row0, nb, Rl);
    {
        const float4* dA4 = reinterpret_cast<const float4*>(dA + row0 * 288);
        const int lim = nb * 72;
        float4 v[FK_IT(FK_TB * 72)];
        float dj[FK_IT(FK_TB * 72)];
#pragma unroll
        for (int it = 0; it < FK_IT(FK_TB * 72); ++it) {   // FK_TB x 72 float4: one (body, joint, row) each
            const int i4 = it * 256 + tid;
            const int ic = i4 < lim ? i4 : 0;
            v[it] = dA4[ic];
            dj[it] = dJp ? dJp[row0 * 72 + ic] : 0.f;  // (body, joint, r) is the same linear index
        }
#pragma unroll
        for (int it = 0; it < FK_IT(FK_TB * 72); ++it) {
            const int i4 = it * 256 + tid;
            if (i4 < lim) {
                const int bdy = i4 / 72, jr = i4 % 72, j = jr / 3, r = jr % 3;
                float* d = D + (j * 12 + r * 4) * FK_GS + bdy;
                d[0] = v[it].x - v[it].w * Js[j * 3];
                d[FK_GS] = v[it].y - v[it].w * Js[j * 3 + 1];
                d[2 * FK_GS] = v[it].z - v[it].w * Js[j * 3 + 2];
                d[3 * FK_GS] = v[it].w + dj[it];
            }
        }
    }
    __syncthreads();
    // ---- phase 2: reverse sweep, 16 lanes per body (one wave holds a body's lanes: LDS program order instead of
    // barriers, as in fk_fwd).  Lane e < 9 = element (r, c): dR_i[r][c] = sum_k Gp[k][r] dG_i[k][c] and the parent's
    // dG_R[p][r][c] += sum_k dG_i[r][k] Ri[c][k] + dgt_i[r] rel_i[c]; lanes 9..11: dG_t[p][r] += dgt_i[r].
    {
        const int bdy = tid >> 4, e = tid & 15;
        if (bdy < nb && e < 12) {
            float* Rr = Rl + bdy * FK_RS;
            const float* Ar = A + (row0 + bdy) * 288;
            const int r = e < 9 ? e / 3 : e - 9, c = e % 3;
#define DL(j, k) D[((j) * 12 + (k)) * FK_GS + bdy]
            // column r of the parent's rotation comes from global memory but does not depend on the sweep: the loads
            // of the next two joints stay in flight (register ring indexed by i & 1; the loop is fully unrolled)
            float Gq[2][3];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int pi = Ps[23 - u];
#pragma unroll
                for (int k = 0; k < 3; ++k) Gq[(23 - u) & 1][k] = Ar[pi * 12 + k * 4 + r];
            }
#pragma unroll
            for (int i = 23; i >= 1; --i) {
                const int p = Ps[i];
                float Gp[3];
#pragma unroll
                for (int k = 0; k < 3; ++k) Gp[k] = Gq[i & 1][k];
                if (i >= 3) {
                    const int pn = Ps[i - 2];
#pragma unroll
                    for (int k = 0; k < 3; ++k) Gq[i & 1][k] = Ar[pn * 12 + k * 4 + r];
                }
                const float dgt = DL(i, r * 4 + 3);
                if (e < 9) {
                    const float dr = Gp[0] * DL(i, c) + Gp[1] * DL(i, 4 + c) + Gp[2] * DL(i, 8 + c);
                    const float up = DL(i, r * 4) * Rr[i * 9 + c * 3] + DL(i, r * 4 + 1) * Rr[i * 9 + c * 3 + 1] +
                                     DL(i, r * 4 + 2) * Rr[i * 9 + c * 3 + 2] + dgt * (Js[i * 3 + c] - Js[p * 3 + c]);
                    // (every lane of the body has read R_i by now -- lockstep -- so dR_i can take its place)
                    Rr[i * 9 + r * 3 + c] = dr;
                    DL(p, r * 4 + c) += up;
                } else {
                    DL(p, r * 4 + 3) += dgt;
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_wave_barrier();
            }
            if (e < 9) Rr[r * 3 + c] = DL(0, r * 4 + c);
#undef DL
        }
    }
    __syncthreads();
    // ---- phase 3: coalesced write-out (+ the pose-feature gradient of joints 1..23)
    {
        const int lim = nb * 216;
        float pfv[FK_IT(FK_TB * 216)];
#pragma unroll
        for (int it = 0; it < FK_IT(FK_TB * 216); ++it) {
            const int idx = it * 256 + tid;
            const int ic = idx < lim ? idx : 0;
            const int bdy = ic / 216, k = ic % 216;
            pfv[it] = (dPF && k >= 9) ? dPF[(row0 + bdy) * lddpf + k - 9] : 0.f;
        }
#pragma unroll
        for (int it = 0; it < FK_IT(FK_TB * 216); ++it) {
            const int idx = it * 256 + tid;
            if (idx < lim) dR[row0 * 216 + idx] = Rl[(idx / 216) * FK_RS + idx % 216] + pfv[it];
        }
    }
}

// ------------------------------------------------------------------------------------------ KP
struct KpArgs {
    long N, V, T;
    const float *A, *Jp, *Mq, *TR, *w0;
    long ldq, ldt;
    const int64_t *view_idx, *frame_idx;
    const float *cams, *targets, *gt_size;
    float focal, cx, cy;
    int add_trans, loss_type, mean_mode;
    const int64_t* n_valid;      // device scalar (or NULL): samples >= *n_valid are padding -- no loss, not counted, no gradient
};

__device__ __forceinline__ int loss_width(int loss_type) { return (loss_type == 2 || loss_type == 3 || loss_type == 5) ? 1 : 2; }

// 3-D position of output joint o of sample s (before the global translation).  Al = the sample's 24
// relative transforms staged in LDS (all lanes of a sample read the same addresses: broadcast).
__device__ __forceinline__ void kp_joint(const KpArgs& a, const KpConst& kc, long s, int o, const float* Al,
                                         float* pos) {
    const int kind = kc.out_kind[o];
    if (kind >= 0) {
        pos[0] = a.Jp[s * 72 + kind * 3]; pos[1] = a.Jp[s * 72 + kind * 3 + 1]; pos[2] = a.Jp[s * 72 + kind * 3 + 2];
        return;
    }
    const int q = -kind - 1;
    const float* M = a.Mq + s * a.ldq + q * 72;
    const float* w = a.w0 + q * 24;
    float p0 = 0.f, p1 = 0.f, p2 = 0.f;
    if ((a.ldq & 3) == 0 && (((uintptr_t)a.Mq) & 15) == 0) {
        // the functional's 72 floats as 18 independent dwordx4 loads (all in flight), not 72 dependent dwords
        float4 m4[18];
#pragma unroll
        for (int i = 0; i < 18; ++i) m4[i] = reinterpret_cast<const float4*>(M)[i];
        const float* m = reinterpret_cast<const float*>(m4);
#pragma unroll
        for (int j = 0; j < 24; ++j) {
            const float m0 = m[j * 3], m1 = m[j * 3 + 1], m2 = m[j * 3 + 2], wj = w[j];
            const float* Aj = Al + j * 12;
            p0 += Aj[0] * m0 + Aj[1] * m1 + Aj[2] * m2 + Aj[3] * wj;
            p1 += Aj[4] * m0 + Aj[5] * m1 + Aj[6] * m2 + Aj[7] * wj;
            p2 += Aj[8] * m0 + Aj[9] * m1 + Aj[10] * m2 + Aj[11] * wj;
        }
    } else {
#pragma unroll 8
        for (int j = 0; j < 24; ++j) {
            const float m0 = M[j * 3], m1 = M[j * 3 + 1], m2 = M[j * 3 + 2], wj = w[j];
            const float* Aj = Al + j * 12;
            p0 += Aj[0] * m0 + Aj[1] * m1 + Aj[2] * m2 + Aj[3] * wj;
            p1 += Aj[4] * m0 + Aj[5] * m1 + Aj[6] * m2 + Aj[7] * wj;
            p2 += Aj[8] * m0 + Aj[9] * m1 + Aj[10] * m2 + Aj[11] * wj;
        }
    }
    pos[0] = p0; pos[1] = p1; pos[2] = p2;
}

// Stage the relative transforms of the block's samples (256/LANES of them) into LDS, coalesced.
template <int LANES>
__device__ __forceinline__ void kp_stage_A(const KpArgs& a, float (*Al)[288]) {
    constexpr int SPB = 256 / LANES;
    const long sbase = (long)blockIdx.x * SPB;
    static_assert(SPB * 288 % 256 == 0, "whole passes of the block");
#pragma unroll
    for (int it = 0; it < SPB * 288 / 256; ++it) {          // (a compile-time trip count: the passes' loads are in flight together)
        const int idx = it * 256 + (int)threadIdx.x;
        const long ss = sbase + idx / 288;
        Al[idx / 288][idx % 288] = ss < a.N ? a.A[ss * 288 + idx % 288] : 0.f;
    }
    __syncthreads();
}

// loss value(s) and d loss / d (u, v) for one joint.  l[2]: per-coordinate losses (W=2) or l[0] (W=1).
__device__ __forceinline__ void kp_loss_eval(int loss_type, float u, float v, float gx, float gy, float conf,
                                             float size, float* l, float* dl_du, float* dl_dv) {
    const float m = conf > 0.5f ? 1.f : 0.f;
    const float rho2 = 10000.f;
    float ru = u - gx, rv = v - gy, k = 1.f;
    if (loss_type == 4) { ru = u / size * 1000.f - gx / size * 1000.f; rv = v / size * 1000.f - gy / size * 1000.f; k = 1000.f / size; }
    if (loss_type == 5) { ru = u / size - gx / size; rv = v / size - gy / size; k = 1.f / size; }
    switch (loss_type) {
        case 0: case 4: {   // mse_robust(_resized): rho^2 r^2 / (r^2 + rho^2) per coordinate
            const float su = ru * ru, sv = rv * rv;
            l[0] = m * (rho2 * (su / (su + rho2))); l[1] = m * (rho2 * (sv / (sv + rho2)));
            const float du_ = su + rho2, dv_ = sv + rho2;
            *dl_du = m * k * 2.f * ru * rho2 * rho2 / (du_ * du_);
            *dl_dv = m * k * 2.f * rv * rho2 * rho2 / (dv_ * dv_);
        } break;
        case 1: {           // mse
            l[0] = m * ru * ru; l[1] = m * rv * rv;
            *dl_du = m * 2.f * ru; *dl_dv = m * 2.f * rv;
        } break;
        case 2: case 5: {   // rmse(_resized): sqrt(1e-6 + |r|^2)
            const float d = sqrtf(1e-6f + (ru * ru + rv * rv));
            l[0] = m * d; l[1] = 0.f;
            *dl_du = m * k * ru / d; *dl_dv = m * k * rv / d;
        } break;
        default: {          // rmse_robust: rho^2 d / (d + rho^2), d = |r|
            const float d = sqrtf(ru * ru + rv * rv);
            l[0] = m * (rho2 * (d / (d + rho2))); l[1] = 0.f;
            const float dd = rho2 * rho2 / ((d + rho2) * (d + rho2));   // d l / d d
            *dl_du = m * dd * ru / d; *dl_dv = m * dd * rv / d;        // NaN at d == 0, as autograd
        }
    }
}

// 32 lanes per sample (n_out <= 32 active), two samples per wave.
__device__ __forceinline__ float group32_sum(float v) {
#pragma unroll
    for (int off = 16; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// Deterministic per-view sums (round 5; common.h): every sample has deposited 12 floats -- [0] its loss x confidence, [1] 1 for a
// real (non-padding) sample, [2 .. 10] its camera gradient -- and the launch's last-arriving block adds, for every view, the
// deposits of the view's samples IN SAMPLE ORDER to view_acc[v][0 .. 1] / d_cams[v][0 .. 8] (components [k0, k1)): the only
// writer of those accumulators, where float atomics used to add per-block partial sums in arrival order.  A batch sorted by
// view (every full batch) is cut into the views' ranges by binary search; any other batch is scanned per view.
constexpr int KP_DEP = 12;
// floats of scratch a launch over N samples of V views takes: the deposits, then the V + 1 view bounds (kp_mark_bounds)
__host__ __device__ inline size_t kp_dep_floats(long N, long V) { return (size_t)N * KP_DEP + (size_t)V + 4; }
// Deterministic per-view sums of the key-point kernels (loss, count, camera gradient).  The kernels DEPOSIT twelve floats per
// sample [loss, count, d cam 0 .. 8, 0] (plain stores) and kp_view_finish_kernel, launched right behind them on the same
// stream, adds a view's samples in a fixed order: one block per view.  (First version: the last-arriving block of the
// depositing launch summed all views -- serial in V: 40 us at 40 views, on the step's main chain.)
// kp_mark_bounds, called for every live sample s (view v) by one lane: ORs "the batch is not sorted by view" into the
// region's second ticket word, and -- for a sorted batch -- leaves the first sample index of every view in bounds[0 .. V]
// (bounds[V] = N): sample s writes the bounds of the views that begin right behind it, sample 0 those up to its own.
__device__ __forceinline__ void kp_mark_bounds(const NemoRed& rr, const int64_t* __restrict__ view_idx, long N, long V, long s, long v) {
    int* bounds = reinterpret_cast<int*>(rr.part + (size_t)N * KP_DEP);
    const long vn = s + 1 < N ? (long)view_idx[s + 1] : V;
    if (vn < v) { __hip_atomic_fetch_or(rr.ticket + 1, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); return; }
    for (long vv = v + 1; vv <= min(vn, V); ++vv) bounds[vv] = (int)(s + 1);
    if (s == 0)
        for (long vv = 0; vv <= min(v, V); ++vv) bounds[vv] = 0;
}
__device__ __forceinline__ void kp_deposit4(float* p, float a, float b, float c, float d) {
    *reinterpret_cast<float4*>(p) = make_float4(a, b, c, d);
}
// Block v: view v's samples -- [bounds[v], bounds[v + 1]) of a sorted batch, otherwise every sample whose view is v -- thread t
// takes samples t, t + 256, ... in ascending order, then the xor-shuffle tree and the four waves in order: a fixed order
// whatever the launch's timing was.  Components [k0, k1) are written: 0, 1 -> view_acc[v][0 .. 1], 2 .. 10 -> d_cams[v][0 .. 8].
// The last block to finish re-arms the region's two ticket words (arrival counter, unsorted flag).
__global__ __launch_bounds__(256) void kp_view_finish_kernel(const float* __restrict__ part, const int64_t* __restrict__ view_idx, long N,
                                                             long V, float* __restrict__ view_acc, float* __restrict__ d_cams, int k0,
                                                             int k1, int* __restrict__ ticket) {
    const long v = blockIdx.x;
    const bool sorted = __hip_atomic_load(ticket + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0;
    const int* bounds = reinterpret_cast<const int*>(part + (size_t)N * KP_DEP);
    const long lo = sorted ? bounds[v] : 0, hi = sorted ? bounds[v + 1] : N;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    float acc[12];
#pragma unroll
    for (int k = 0; k < 12; ++k) acc[k] = 0.f;
    const float4* p4 = reinterpret_cast<const float4*>(part);
    for (long s0 = lo + threadIdx.x; s0 < hi; s0 += 1024) {           // four samples per thread: their twelve 16-byte loads in flight
        float4 q[4][3];
        bool on[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            // (unconditional loads from a clamped index: behind a predicate per load hipcc branches around each and waits for it -- one
            //  round trip per load instead of one per pass)
            const long s = min(s0 + 256 * u, hi - 1);
            on[u] = s0 + 256 * u < hi && (sorted || view_idx[s] == v);
#pragma unroll
            for (int c = 0; c < 3; ++c) q[u][c] = p4[s * 3 + c];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (!on[u]) continue;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                acc[4 * c] += q[u][c].x; acc[4 * c + 1] += q[u][c].y; acc[4 * c + 2] += q[u][c].z; acc[4 * c + 3] += q[u][c].w;
            }
        }
    }
    __shared__ float wsum4[4][12];
#pragma unroll
    for (int k = 0; k < 11; ++k) {
        float r = acc[k];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) r += __shfl_xor(r, off, 64);
        if (lane == 0) wsum4[wv][k] = r;
    }
    __syncthreads();
    if (threadIdx.x < 11 && (int)threadIdx.x >= k0 && (int)threadIdx.x < k1) {
        const int k = threadIdx.x;
        const float mine = (wsum4[0][k] + wsum4[1][k]) + (wsum4[2][k] + wsum4[3][k]);
        if (k < 2) { if (view_acc) view_acc[v * 2 + k] += mine; }
        else if (mine != 0.f) d_cams[v * 9 + (k - 2)] += mine;
    }
    if (threadIdx.x == 0 && __hip_atomic_fetch_add(ticket, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (int)gridDim.x - 1) {
        __hip_atomic_store(ticket, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(ticket + 1, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}
static void kp_launch_view_finish(const NemoRed& rr, const int64_t* view_idx, long N, long V, float* view_acc, float* d_cams, int k0, int k1,
                                  hipStream_t st) {
    if (!rr.part || V <= 0) return;
    hipLaunchKernelGGL(kp_view_finish_kernel, dim3((unsigned)V), dim3(256), 0, st, rr.part, view_idx, N, V, view_acc, d_cams, k0, k1,
                       rr.ticket);
}

template <int LANES>
__global__ __launch_bounds__(256) void kp_fwd_kernel(KpArgs a, KpConst kc, float* __restrict__ j3d,
                                                     float* __restrict__ p2d, float* __restrict__ loss_all,
                                                     float* __restrict__ view_acc, NemoRed rr) {
    const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long s = t / LANES;
    const int o = (int)(t % LANES);
    const bool active = s < a.N && o < kc.n_out;
    __shared__ float Al[256 / LANES][288];
    kp_stage_A<LANES>(a, Al);
    float wsum = 0.f;
    long v = 0;
    if (s < a.N) v = a.view_idx[s];
    const bool pad = a.n_valid && s >= *a.n_valid;
    if (active) {
        float pos[3];
        kp_joint(a, kc, s, o, Al[threadIdx.x / LANES], pos);
        if (a.add_trans) {
#pragma unroll
            for (int c = 0; c < 3; ++c) pos[c] += a.TR[s * a.ldt + c] - a.TR[a.N * a.ldt + c];
        }
        if (j3d) { j3d[(s * kc.n_out + o) * 3] = pos[0]; j3d[(s * kc.n_out + o) * 3 + 1] = pos[1]; j3d[(s * kc.n_out + o) * 3 + 2] = pos[2]; }
        const float* cam = a.cams + v * 9;
        float Rc[9];
        rot6d_fwd(cam + 3, Rc);
        const float px = Rc[0] * pos[0] + Rc[1] * pos[1] + Rc[2] * pos[2] + cam[0];
        const float py = Rc[3] * pos[0] + Rc[4] * pos[1] + Rc[5] * pos[2] + cam[1];
        const float pz = Rc[6] * pos[0] + Rc[7] * pos[1] + Rc[8] * pos[2] + cam[2];
        const float nx = px / pz, ny = py / pz, nz = pz / pz;
        const float u = a.focal * nx + a.cx * nz, w = a.focal * ny + a.cy * nz;
        if (p2d) { p2d[(s * kc.n_out + o) * 2] = u; p2d[(s * kc.n_out + o) * 2 + 1] = w; }
        if (a.targets) {
            const long f = a.frame_idx[s];
            const float* g = a.targets + ((v * a.T + f) * kc.n_out + o) * 3;
            const float size = a.gt_size ? a.gt_size[v * a.T + f] : 1.f;
            float l[2], du, dv;
            kp_loss_eval(a.loss_type, u, w, g[0], g[1], g[2], size, l, &du, &dv);
            if (pad) { l[0] = 0.f; l[1] = 0.f; }
            const int W = loss_width(a.loss_type);
            if (loss_all) {
                loss_all[(s * kc.n_out + o) * W] = l[0];
                if (W == 2) loss_all[(s * kc.n_out + o) * W + 1] = l[1];
            }
            wsum = (l[0] + (W == 2 ? l[1] : 0.f)) * (a.mean_mode == 0 ? g[2] : 1.f);
        }
    }
    if (view_acc && rr.part) {
        // deterministic: per-sample deposits, summed per view in sample order by kp_view_finish_kernel
        if (LANES == 32) wsum = group32_sum(wsum);
        if (o == 0 && s < a.N) {
            kp_deposit4(rr.part + s * KP_DEP, pad ? 0.f : wsum, pad ? 0.f : 1.f, 0.f, 0.f);
            kp_mark_bounds(rr, a.view_idx, a.N, a.V, s, v);
        }
    } else if (view_acc) {
        // one atomic per (block, view) instead of one per sample: the V x 2 accumulators are hot
        // same-address targets (300 serialised L2 atomics each at N = 2400 otherwise)
        constexpr int SPB = 256 / LANES;
        __shared__ float bsum[SPB];
        __shared__ long bview[SPB];
        if (LANES == 32) wsum = group32_sum(wsum);
        const int sl = threadIdx.x / LANES;
        if (o == 0) { bsum[sl] = wsum; bview[sl] = (s < a.N && !pad) ? v : -1; }
        __syncthreads();
        if (threadIdx.x < SPB && bview[threadIdx.x] >= 0) {
            const long mv = bview[threadIdx.x];
            bool first = true;
            for (int k = 0; k < (int)threadIdx.x; ++k) first = first && bview[k] != mv;
            if (first) {
                float tot = 0.f, cnt = 0.f;
                for (int k = threadIdx.x; k < SPB; ++k)
                    if (bview[k] == mv) { tot += bsum[k]; cnt += 1.f; }
                atomicAdd(view_acc + mv * 2, tot);
                atomicAdd(view_acc + mv * 2 + 1, cnt);
            }
        }
    }
}

__global__ void kp_finalize_kernel(long V, int n_out, int W, int mean_mode, const float* __restrict__ view_acc,
                                   float* __restrict__ scalar_out, float* __restrict__ norm) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    float total = 0.f, nu = 0.f, cnt = 0.f, sum = 0.f;
    for (long v = 0; v < V; ++v) {
        const float c = view_acc[v * 2 + 1];
        if (c > 0.f) { total += view_acc[v * 2] / (c * (float)(n_out * W)); nu += 1.f; }
        cnt += c; sum += view_acc[v * 2];
    }
    if (mean_mode == 0) { *scalar_out += nu > 0.f ? total / nu : 0.f; norm[0] = nu; }
    else { *scalar_out += cnt > 0.f ? sum / (cnt * (float)(n_out * W)) : 0.f; norm[0] = cnt; }
}

// FUSED (nemo_kp_fwd_bwd): the forward of kp_fwd_kernel and this backward in ONE launch.  The only thing the backward needs
// from a completed forward is the per-view normaliser, and that is a function of the INDICES alone (view_cnt[v] = samples of
// view v in the batch) -- the caller passes it, the launch writes the forward's outputs (j3d, p2d, loss_all, view_acc) on the
// way and needs no predecessor launch: one dependent node and one pass over the joints less on the step's main chain.
template <int LANES, bool FUSED = false>
__global__ __launch_bounds__(256) void kp_bwd_kernel(KpArgs a, KpConst kc, const float* __restrict__ view_acc,
                                                     const float* __restrict__ norm, float upstream,
                                                     float* __restrict__ dA, float* __restrict__ dJp,
                                                     float* __restrict__ dMq, float* __restrict__ dTR,
                                                     long lddt, float* __restrict__ d_cams, int nq,
                                                     const float* __restrict__ dj3d_extra,
                                                     const int64_t* __restrict__ view_cnt = nullptr,
                                                     float* __restrict__ j3d = nullptr, float* __restrict__ p2d = nullptr,
                                                     float* __restrict__ loss_all = nullptr,
                                                     float* __restrict__ view_acc_out = nullptr,
                                                     NemoRed rr = NemoRed{nullptr, nullptr}) {
    const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long s = t / LANES;
    const int o = (int)(t % LANES);
    const bool live = s < a.N;
    const bool active = live && o < kc.n_out;
    __shared__ float Al[256 / LANES][288];
    kp_stage_A<LANES>(a, Al);
    // norm == NULL: the normaliser nemo_kp_finalize would write (number of views present / total confidence count) is
    // derived here from the per-view accumulators, so that kp_finalize is off the dependency chain of the step
    float nrm;
    if (!FUSED && norm) {
        nrm = norm[0];
    } else {
        __shared__ float nred[4];
        float part = 0.f;
        for (long vv = threadIdx.x; vv < a.V; vv += 256) {
            const float c = FUSED ? (float)view_cnt[vv] : view_acc[vv * 2 + 1];
            part += a.mean_mode == 0 ? (c > 0.f ? 1.f : 0.f) : c;
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) part += __shfl_xor(part, off, 64);
        if ((threadIdx.x & 63) == 0) nred[threadIdx.x >> 6] = part;
        __syncthreads();
        nrm = nred[0] + nred[1] + nred[2] + nred[3];
    }
    long v = 0;
    if (live) v = a.view_idx[s];
    const bool pad = a.n_valid && s >= *a.n_valid;
    float dpos[3] = {0.f, 0.f, 0.f};
    float dcam[9] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    int kind = 0;
    float wsum = 0.f;                // FUSED: this lane's contribution to the view's loss accumulator
    if (active) kind = kc.out_kind[o];
    if (active && (!pad || FUSED)) {
        float pos[3];
        kp_joint(a, kc, s, o, Al[threadIdx.x / LANES], pos);
        if (a.add_trans) {
#pragma unroll
            for (int c = 0; c < 3; ++c) pos[c] += a.TR[s * a.ldt + c] - a.TR[a.N * a.ldt + c];
        }
        if (FUSED && j3d) { j3d[(s * kc.n_out + o) * 3] = pos[0]; j3d[(s * kc.n_out + o) * 3 + 1] = pos[1]; j3d[(s * kc.n_out + o) * 3 + 2] = pos[2]; }
        const float* cam = a.cams + v * 9;
        float Rc[9];
        rot6d_fwd(cam + 3, Rc);
        const float px = Rc[0] * pos[0] + Rc[1] * pos[1] + Rc[2] * pos[2] + cam[0];
        const float py = Rc[3] * pos[0] + Rc[4] * pos[1] + Rc[5] * pos[2] + cam[1];
        const float pz = Rc[6] * pos[0] + Rc[7] * pos[1] + Rc[8] * pos[2] + cam[2];
        const float nx = px / pz, ny = py / pz, nz = pz / pz;
        const float u = a.focal * nx + a.cx * nz, w = a.focal * ny + a.cy * nz;
        if (FUSED && p2d) { p2d[(s * kc.n_out + o) * 2] = u; p2d[(s * kc.n_out + o) * 2 + 1] = w; }
        const long f = a.frame_idx[s];
        const float* g = a.targets + ((v * a.T + f) * kc.n_out + o) * 3;
        const float size = a.gt_size ? a.gt_size[v * a.T + f] : 1.f;
        float l[2], du, dv;
        kp_loss_eval(a.loss_type, u, w, g[0], g[1], g[2], size, l, &du, &dv);
        const int W = loss_width(a.loss_type);
        if (FUSED) {
            if (pad) { l[0] = 0.f; l[1] = 0.f; du = 0.f; dv = 0.f; }
            if (loss_all) {
                loss_all[(s * kc.n_out + o) * W] = l[0];
                if (W == 2) loss_all[(s * kc.n_out + o) * W + 1] = l[1];
            }
            wsum = (l[0] + (W == 2 ? l[1] : 0.f)) * (a.mean_mode == 0 ? g[2] : 1.f);
        }
        // d total / d loss_all element
        const float cnt_v = FUSED ? (float)view_cnt[v] : view_acc[v * 2 + 1];
        float coef;
        if (a.mean_mode == 0) coef = upstream * g[2] / (nrm * cnt_v * (float)(kc.n_out * W));
        else coef = upstream / (nrm * (float)(kc.n_out * W));
        if (FUSED && pad) coef = 0.f;
        du *= coef; dv *= coef;
        // u = f*px/pz + cx*(pz/pz)
        const float dpx = du * a.focal / pz, dpy = dv * a.focal / pz;
        const float dpz = -(du * a.focal * px + dv * a.focal * py) / (pz * pz);
        dpos[0] = Rc[0] * dpx + Rc[3] * dpy + Rc[6] * dpz;
        dpos[1] = Rc[1] * dpx + Rc[4] * dpy + Rc[7] * dpz;
        dpos[2] = Rc[2] * dpx + Rc[5] * dpy + Rc[8] * dpz;
        const float dRc[9] = {dpx * pos[0], dpx * pos[1], dpx * pos[2], dpy * pos[0], dpy * pos[1],
                              dpy * pos[2], dpz * pos[0], dpz * pos[1], dpz * pos[2]};
        dcam[0] = dpx; dcam[1] = dpy; dcam[2] = dpz;
        rot6d_bwd(cam + 3, dRc, dcam + 3);
        if (dj3d_extra) {       // gradient of a world-space term on the output joints (temporal smoothness)
            const float* x = dj3d_extra + (s * kc.n_out + o) * 3;
            dpos[0] += x[0]; dpos[1] += x[1]; dpos[2] += x[2];
        }
    }
    const bool ordered = rr.part != nullptr && ((FUSED && view_acc_out) || d_cams);
    if (ordered) {
        // deterministic per-view sums (kp_view_finish): this sample's loss, count and camera gradient as one deposit
        float dep[11];
        dep[0] = LANES == 32 ? group32_sum(wsum) : wsum;
        dep[1] = (live && !pad) ? 1.f : 0.f;
#pragma unroll
        for (int k = 0; k < 9; ++k) dep[2 + k] = LANES == 32 ? group32_sum(dcam[k]) : dcam[k];
        if (o < 3 && live) {                               // (every lane of the group holds all sums: lanes 0 .. 2 store 16 bytes each)
            const float q0 = o == 0 ? dep[0] : (o == 1 ? dep[4] : dep[8]), q1 = o == 0 ? dep[1] : (o == 1 ? dep[5] : dep[9]);
            const float q2 = o == 0 ? dep[2] : (o == 1 ? dep[6] : dep[10]), q3 = o == 0 ? dep[3] : (o == 1 ? dep[7] : 0.f);
            kp_deposit4(rr.part + s * KP_DEP + 4 * o, q0, q1, q2, q3);
            if (o == 0) kp_mark_bounds(rr, a.view_idx, a.N, a.V, s, v);
        }
    }
    if (!ordered && FUSED && view_acc_out) {
        // the forward's per-view accumulators [sum(loss * conf), #samples]: one atomic per (block, view), as in kp_fwd_kernel
        constexpr int SPBf = 256 / LANES;
        __shared__ float fsum[SPBf];
        __shared__ long fview[SPBf];
        if (LANES == 32) wsum = group32_sum(wsum);
        const int slf = threadIdx.x / LANES;
        if (o == 0) { fsum[slf] = wsum; fview[slf] = (live && !pad) ? v : -1; }
        __syncthreads();
        if (threadIdx.x < SPBf && fview[threadIdx.x] >= 0) {
            const long mv = fview[threadIdx.x];
            bool first = true;
            for (int k = 0; k < (int)threadIdx.x; ++k) first = first && fview[k] != mv;
            if (first) {
                float tot = 0.f, cnt = 0.f;
                for (int k = threadIdx.x; k < SPBf; ++k)
                    if (fview[k] == mv) { tot += fsum[k]; cnt += 1.f; }
                atomicAdd(view_acc_out + mv * 2, tot);
                atomicAdd(view_acc_out + mv * 2 + 1, cnt);
            }
        }
    }
    // camera gradient: reduce over the sample's joints (shuffles), then over the block's samples of the
    // same view (LDS), one atomic per (block, view, component)
    if (!ordered && d_cams) {
        constexpr int SPBc = 256 / LANES;
        __shared__ float bcam[SPBc][9];
        __shared__ long bviewc[SPBc];
        const int slc = threadIdx.x / LANES;
#pragma unroll
        for (int k = 0; k < 9; ++k) {
            float r = dcam[k];
            if (LANES == 32) r = group32_sum(r);
            if (o == 0) bcam[slc][k] = r;
        }
        if (o == 0) bviewc[slc] = live ? v : -1;
        __syncthreads();
        if (threadIdx.x < SPBc * 9) {
            const int sl2 = threadIdx.x / 9, k = threadIdx.x % 9;
            const long mv = bviewc[sl2];
            if (mv >= 0) {
                bool first = true;
                for (int q2 = 0; q2 < sl2; ++q2) first = first && bviewc[q2] != mv;
                if (first) {
                    float tot = 0.f;
                    for (int q2 = sl2; q2 < SPBc; ++q2)
                        if (bviewc[q2] == mv) tot += bcam[q2][k];
                    if (tot != 0.f) atomicAdd(d_cams + mv * 9 + k, tot);
                }
            }
        }
    }
    // translation gradient
    if (dTR && a.add_trans) {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            float r = dpos[c];
            if (LANES == 32) r = group32_sum(r);
            if (live && o == 0) dTR[s * lddt + c] = r;
        }
    }
    if (dA) {
    // kinematic joints: dJp[s][j] += the position gradients of the output joints that ARE kinematic joint j, added in output
    // order by ONE thread per (sample, joint, component) (round 5: float atomics per output joint before -- several outputs can
    // share a joint, and the order of their additions changed from run to run)
    // (the outputs of a joint as a list built once per block: a loop over all outputs with a look-up and a branch per
    //  iteration was 75 dependent LDS round trips per thread, 8 us of the 8 x 300 step's kernel)
    __shared__ float dkin[256 / LANES][NEMO_MAX_OUT][3];
    __shared__ unsigned char klist[24][NEMO_MAX_OUT];
    __shared__ int kcnt[24];
    if (threadIdx.x < 24) {
        int n = 0;
        for (int q = 0; q < kc.n_out && q < NEMO_MAX_OUT; ++q)
            if (kc.out_kind[q] == (int)threadIdx.x) klist[threadIdx.x][n++] = (unsigned char)q;
        kcnt[threadIdx.x] = n;
    }
    if (o < kc.n_out && o < NEMO_MAX_OUT) {
        const int slk = threadIdx.x / LANES;
        dkin[slk][o][0] = active ? dpos[0] : 0.f; dkin[slk][o][1] = active ? dpos[1] : 0.f; dkin[slk][o][2] = active ? dpos[2] : 0.f;
    }
    __syncthreads();
    for (int idx = threadIdx.x; idx < (256 / LANES) * 72; idx += 256) {
        const int ls = idx / 72, j = (idx % 72) / 3, c = idx % 3;
        const long ss = (long)blockIdx.x * (256 / LANES) + ls;
        const int n = kcnt[j];
        if (ss >= a.N || n == 0) continue;
        float acc = 0.f;
        for (int i = 0; i < n; ++i) acc += dkin[ls][klist[j][i]][c];
        dJp[ss * 72 + j * 3 + c] += acc;
    }
    // mesh functionals: pos = sum_j A_R[j] Mq[q][j] + A_t[j] w0[q][j].
    // d Mq[q][j] = A_R[j]^T dpos_q is private to the (sample, joint) lane; dA[j] sums over the mesh
    // joints of the sample: stage dpos in LDS and let every thread own output entries of dA.
    __shared__ float dps[256 / LANES][NEMO_MAX_OUT][3];      // indexed by mesh functional q
    const int sl = threadIdx.x / LANES;
    const bool mesh = active && kind < 0;
    if (mesh) {
        const int q = -kind - 1;
        dps[sl][q][0] = dpos[0]; dps[sl][q][1] = dpos[1]; dps[sl][q][2] = dpos[2];
#pragma unroll 8
        for (int j = 0; j < 24; ++j) {
            const float* Aj = Al[sl] + j * 12;
            float* dM = dMq + s * a.ldq + q * 72 + j * 3;
            dM[0] = Aj[0] * dpos[0] + Aj[4] * dpos[1] + Aj[8] * dpos[2];
            dM[1] = Aj[1] * dpos[0] + Aj[5] * dpos[1] + Aj[9] * dpos[2];
            dM[2] = Aj[2] * dpos[0] + Aj[6] * dpos[1] + Aj[10] * dpos[2];
        }
    }
    __syncthreads();
    // dA[j][r][c] = sum_q dpos_q[r] * (c < 3 ? Mq[q][j][c] : w0[q][j]): a thread owns (sample, j, c) and its
    // three rows r; the nq operand loads are independent of each other (unrolled: all in flight at once --
    // a loop over the output joints with a look-up and a branch per iteration cost 32 us of dependent
    // L2 round trips)
    constexpr int SPB = 256 / LANES;                  // samples per block
    const long sbase = (long)blockIdx.x * SPB;
    for (int idx = threadIdx.x; idx < SPB * 96; idx += 256) {
        const int ls = idx / 96, jc = idx % 96, j = jc / 4, c = jc % 4;
        const long ss = sbase + ls;
        if (ss >= a.N) continue;
        const float* Ms = c < 3 ? a.Mq + ss * a.ldq + j * 3 + c : a.w0 + j;
        const int qs = c < 3 ? 72 : 24;
        float a0 = 0.f, a1 = 0.f, a2 = 0.f;
#pragma unroll 4
        for (int q = 0; q < nq; ++q) {
            const float m = Ms[q * qs];
            a0 += dps[ls][q][0] * m; a1 += dps[ls][q][1] * m; a2 += dps[ls][q][2] * m;
        }
        float* o3 = dA + ss * 288 + j * 12 + c;
        o3[0] = a0; o3[4] = a1; o3[8] = a2;
    }
    }
}

__global__ __launch_bounds__(256) void project_kernel(long N, int Jn, const float* __restrict__ pts,
                                                      const int64_t* __restrict__ view_idx,
                                                      const float* __restrict__ cams, float focal, float cx,
                                                      float cy, float* __restrict__ p2d) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N * Jn) return;
    const long s = i / Jn;
    const float* cam = cams + view_idx[s] * 9;
    float Rc[9];
    rot6d_fwd(cam + 3, Rc);
    const float x = pts[i * 3], y = pts[i * 3 + 1], z = pts[i * 3 + 2];
    const float px = Rc[0] * x + Rc[1] * y + Rc[2] * z + cam[0];
    const float py = Rc[3] * x + Rc[4] * y + Rc[5] * z + cam[1];
    const float pz = Rc[6] * x + Rc[7] * y + Rc[8] * z + cam[2];
    const float nz = pz / pz;
    p2d[i * 2] = focal * (px / pz) + cx * nz;
    p2d[i * 2 + 1] = focal * (py / pz) + cy * nz;
}

// ------------------------------------------------------------------------------------------ mesh
// verts[row][v] = T[row][v] * [VP[row][v]; 1] (+ trans[row]),  T = sum_j W[v][j] A[row][j]
__global__ __launch_bounds__(256) void skin_vertices_kernel(long rows, long NV, const float* __restrict__ VP,
                                                            long ldvp, const float* __restrict__ A,
                                                            const float* __restrict__ Wt,
                                                            const float* __restrict__ trans, long ldt,
                                                            float* __restrict__ verts) {
    const long v = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const bool valid = v < NV;
    float w[24];
#pragma unroll
    for (int j = 0; j < 24; ++j) w[j] = valid ? Wt[j * NV + v] : 0.f;
    for (long row = blockIdx.y; row < rows; row += gridDim.y) {
        const float* Ar = A + row * 288;     // block-uniform -> scalar loads
        float T[12];
#pragma unroll
        for (int e = 0; e < 12; ++e) T[e] = 0.f;
#pragma unroll
        for (int j = 0; j < 24; ++j)
#pragma unroll
            for (int e = 0; e < 12; ++e) T[e] += w[j] * Ar[j * 12 + e];
        if (valid) {
            const float* vp = VP + row * ldvp + v * 3;
            const float x = vp[0], y = vp[1], z = vp[2];
            float* out = verts + (row * NV + v) * 3;
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                float o = T[r * 4] * x + T[r * 4 + 1] * y + T[r * 4 + 2] * z + T[r * 4 + 3];
                if (trans) o += trans[row * ldt + r];
                out[r] = o;
            }
        }
    }
}

// VPoser v2v term.  Block = S bodies (orig row s, reconstruction row N+s) x all vertices in chunks of
// 256.  Forward skinning on the VALU: the per-vertex weights sit in LDS (stride 25, conflict-free),
// the block-uniform 3x4 transforms are streamed joint by joint through SGPRs (a runtime j loop keeps
// the scalar live ranges short -- fully unrolling it made the compiler spill ~1700 SGPRs into VGPR
// lanes).  d(sum|.|)/dA is reduced over the vertices on the matrix cores:
// dA[j][e] = sum_v W[v][j] dT[v][e]  (16x16x4 f32 MFMA, K = vertices).
template <int S>
__global__ __launch_bounds__(256) void v2v_skin_l1_kernel(long N, long NV, const float* __restrict__ VP,
                                                          long ldvp, const float* __restrict__ A,
                                                          const float* __restrict__ Wt,
                                                          float* __restrict__ loss_sum,
                                                          float* __restrict__ dVP, long lddvp,
                                                          float* __restrict__ dA) {
    __shared__ float Wl[256][25];        // chunk weights, stride 25 -> conflict-free row reads
    __shared__ float dTl[4][64][13];     // per-wave dT tile
    __shared__ float red[16];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int l15 = lane & 15, lq = lane >> 4;
    const long s0 = (long)blockIdx.x * S;

    f32x4 acc[S][2];
#pragma unroll
    for (int si = 0; si < S; ++si)
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[si][t][r] = 0.f;
    float lsum = 0.f;

    for (long c0 = 0; c0 < NV; c0 += 256) {
        const long v = c0 + tid;
        const bool valid = v < NV;
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 24; ++j) Wl[tid][j] = valid ? Wt[j * NV + v] : 0.f;
        __syncthreads();
        // A-operand fragments: Aop[i = joint][k = vertex] = W[vertex][joint]
        float af[16][2];
#pragma unroll
        for (int kk = 0; kk < 16; ++kk) {
            const int vr = wid * 64 + 4 * kk + lq;
            af[kk][0] = Wl[vr][l15];
            af[kk][1] = (l15 < 8) ? Wl[vr][16 + l15] : 0.f;
        }
#pragma unroll
        for (int si = 0; si < S; ++si) {
            const long s = s0 + si;
            if (s >= N) break;                 // block-uniform
            const float* Ao = A + s * 288;
            const float* Ar = A + (N + s) * 288;
            float To[12], Tr[12];
#pragma unroll
            for (int e = 0; e < 12; ++e) { To[e] = 0.f; Tr[e] = 0.f; }
#pragma unroll 2
            for (int j = 0; j < 24; ++j) {
                const float wj = Wl[tid][j];
#pragma unroll
                for (int e = 0; e < 12; ++e) {
                    To[e] += wj * Ao[j * 12 + e];
                    Tr[e] += wj * Ar[j * 12 + e];
                }
            }
            float po[3] = {0.f, 0.f, 0.f}, pr[3] = {0.f, 0.f, 0.f};
            if (valid) {
                const float* a = VP + s * ldvp + v * 3;
                const float* b = VP + (N + s) * ldvp + v * 3;
                po[0] = a[0]; po[1] = a[1]; po[2] = a[2];
                pr[0] = b[0]; pr[1] = b[1]; pr[2] = b[2];
            }
            float g[3];
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                const float vo = To[r * 4] * po[0] + To[r * 4 + 1] * po[1] + To[r * 4 + 2] * po[2] + To[r * 4 + 3];
                const float vr = Tr[r * 4] * pr[0] + Tr[r * 4 + 1] * pr[1] + Tr[r * 4 + 2] * pr[2] + Tr[r * 4 + 3];
                const float d = vr - vo;
                lsum += valid ? fabsf(d) : 0.f;
                // d |v_rec - v_orig| / d v_orig = -sign(d)
                g[r] = (!valid || d == 0.f) ? 0.f : (d > 0.f ? -1.f : 1.f);
            }
            if (valid) {
                float* o = dVP + s * lddvp + v * 3;
                o[0] = To[0] * g[0] + To[4] * g[1] + To[8] * g[2];
                o[1] = To[1] * g[0] + To[5] * g[1] + To[9] * g[2];
                o[2] = To[2] * g[0] + To[6] * g[1] + To[10] * g[2];
            }
            __syncthreads();                   // previous body's tile fully consumed
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                dTl[wid][lane][r * 4 + 0] = g[r] * po[0];
                dTl[wid][lane][r * 4 + 1] = g[r] * po[1];
                dTl[wid][lane][r * 4 + 2] = g[r] * po[2];
                dTl[wid][lane][r * 4 + 3] = g[r];
            }
            __syncthreads();
#pragma unroll
            for (int kk = 0; kk < 16; ++kk) {
                const float b = (l15 < 12) ? dTl[wid][4 * kk + lq][l15] : 0.f;
                acc[si][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[kk][0], b, acc[si][0], 0, 0, 0);
                acc[si][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[kk][1], b, acc[si][1], 0, 0, 0);
            }
        }
    }
    // cross-wave reduction of dA through LDS (reuse Wl as scratch: 4 waves x 32 joints x 16 cols)
    __syncthreads();
    float* scratch = &Wl[0][0];                // 6400 floats available, need 4*512
#pragma unroll
    for (int si = 0; si < S; ++si) {
        const long s = s0 + si;
        if (s >= N) break;
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) scratch[wid * 512 + (t * 16 + lq * 4 + r) * 16 + l15] = acc[si][t][r];
        __syncthreads();
        for (int idx = tid; idx < 288; idx += 256) {
            const int j = idx / 12, e = idx % 12;
            const int off = j * 16 + e;
            dA[s * 288 + idx] = scratch[off] + scratch[512 + off] + scratch[1024 + off] + scratch[1536 + off];
        }
        __syncthreads();
    }
    const float tot = block_sum(lsum, red);
    if (tid == 0) atomicAdd(loss_sum, tot);
}

// ------------------------------------------------------------------------------------------
// Fused full-mesh v2v term: pose blend + skinning of BOTH bodies + L1 + its gradient, entirely in the
// 16x16 MFMA accumulator layout (v_mfma_f32_16x16x4_f32; rows i = 16 vertices, columns n = 16
// samples).  For one (vertex tile, sample group) a lane owns 4 vertices x 1 sample of
//   vp_c[v][s]  = v_shaped + sum_p P[p][3v+c] pf[s][p]        (K = 207, A-operand streamed from L2)
//   T_e[v][s]   = sum_j W[v][j] A[s][j][e]                     (K = 24, 12 entries e of the 3x4 map)
// so vert = T.[vp;1], the |v_rec - v_orig| term, g = -sign, dvp = T^T g and dT = g (x) [vp;1] are all
// lane-local VALU work, and dT is *already* the B-operand (k = vertex) of the vertex->joint reduction
// dA[j][e][s] += sum_v W[v][j] dT_e[v][s].  Nothing of the 2N x 20670 blended mesh ever touches HBM;
// only dVP^T (3NV x N, the operand of the blend-shape adjoint GEMM) is written.
// Block = one group of 16 samples (their pose features and transforms for both bodies staged once in
// LDS, strides == 2 mod 32 -> conflict-free B-operand reads) x one vertex range; the 4 waves take
// alternating vertex tiles.
#define MF_PFS 226      // pose-feature row stride in LDS (>= 208, == 2 mod 32)
#define MF_AS 290       // transform row stride in LDS   (>= 288, == 2 mod 32)
#define MF_PFB 232      // bf16 variant: pose-feature row stride in bf16 elements (464 B: the 16 rows a ds_read_b128 lane
                        // group touches land in 16 different 16-byte bank quads)
typedef __bf16 mbf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4m __attribute__((ext_vector_type(4)));
typedef _Float16 mf16x8 __attribute__((ext_vector_type(8)));

// MODE 1 / 2 (bf16): the pose blend (K = 207, 57 % of the kernel's MFMAs) runs on v_mfma_f32_16x16x32_bf16 -- blend shapes
// rounded to bf16 once in nemo_ctx_create, pose features rounded when they are staged, fp32 accumulate.  BASELINE
// configs[2].  P then points at nemo_ctx::d_posedirs_bf16 and ldP is NVp.
//   MODE 1 (round 2): skinning, L1 and both adjoints on the fp32 pipe.
//   MODE 3: the vertex->joint adjoint on the bf16 pipe in SPLIT precision (its operands are lane-local), skinning on the fp32
//   pipe (dense weights) or as sparse VALU FMAs (SPARSE).  MODE 2 (the bf16 default since round 5): skinning in split precision
//   too.  Round 3 measured MODE 2 SLOWER (475 against 358 us at 8 x 300) -- but with every ds_read_b128 of the kernel 8-byte
//   aligned (64 LDS cycles instead of 4, fixed in round 4); re-measured it is 4 - 7 % faster than MODE 3 with sparse skinning
//   (434 against 465 us per 40 x 300 launch): the bf16 kernel is VALU-issue bound and the 384 skinning FMAs per tile leave the
//   VALU for 72 MFMAs on a matrix pipe that is 23 % busy (profiles/r05_pmc_mesh_b16.md).
//   Split precision: every fp32 operand is
//   carried as two bf16 pieces (hi = bf16(x), lo = bf16(x - hi): 16 significant bits, ~4e-6 relative) and the product
//   is the sum of the piece products with fp32 accumulation:
//     T_e = W A_e        : hi*hi + hi*lo + lo*hi, K = 24 joints in one K = 32 instruction  (36 MFMAs per body, was 72)
//     dA_e = W^T dT_e    : K = 16 vertices x 2 pieces fill one K = 32 instruction: [Whi | Whi] x [dThi ; dTlo] +
//                          [Wlo | Wlo] x [dThi ; dTlo] = all four piece products         (48 MFMAs, was 96)
//   at 16 cycles per instruction instead of 32: 2592 MFMA cycles per vertex tile against 8352 in MODE 1.  The error of
//   the split (1e-5) is far below the bf16 blend's (4e-3 of the pose offsets), so MODE 2 meets MODE 1's tolerances.
#ifndef MESH_PQD
#define MESH_PQD 3
#endif
#define MF_ASP 288      // SPARSE: transforms in LDS as [body][joint][row c][sample 16][4 entries d] -- (joint, row) segments of 256
                        // bytes, sample-major: a ds_read_b128 whose lanes read 16 bytes at 16 * (lane & 15) of ANY four
                        // segments (one per 16-lane quarter: the quarters' vertices have different joints) is the access
                        // the LDS serves at full rate (4 cycles).  A row-per-sample layout (lane stride 1168 B) measured
                        // 66 LDS cycles per ds_read_b128 -- one lane per cycle, LDS 100 % busy, SQ_LDS_BANK_CONFLICT ~ 0.
#define MF_TAIL 32      // block-reduction scratch + flags at the end of the dynamic LDS block (floats)
#define MF_AB 392       // MODE 2: transform row stride in bf16 elements (12 entries x 32 joints + 8: 784 B, the 16 sample
                        // rows of a ds_read_b128 lane group land in 16 different 16-byte bank quads)
// Sparse skinning of one output row (4 transform entries) for a lane's 4 vertices x 1 sample: T4[d][r] = sum_q w[r][q] *
// A[joint(r, q)][4 c + d].  Ab = the lane's sample column of row c of joint 0 in LDS (layout: MF_ASP); sj = 3 * joint per
// byte, i.e. the joint's segment index.  4 RB float4 reads are issued BEFORE their FMAs.  (Scalar v_fmac_f32 on purpose:
// the same sums as v_pk_fma_f32 pairs measured 555 against 494 us per launch at 8 x 300 -- packed fp32 VALU beside
// MFMAs is the slower form on this chip, see the Makefile.)
template <int RB>      // RB = vertices (of the lane's 4) per batch: 4 RB reads in flight
__device__ __forceinline__ void sparse_rows(const float* Ab, const float4 (&sw)[4], const unsigned int (&sj)[4], f32x4 (&T4)[4]) {
#pragma unroll
    for (int h = 0; h < 4 / RB; ++h) {
        float4 a[RB][4];
#pragma unroll
        for (int rr = 0; rr < RB; ++rr)
#pragma unroll
            for (int q = 0; q < 4; ++q)
                a[rr][q] = *reinterpret_cast<const float4*>(Ab + 64 * ((sj[RB * h + rr] >> (8 * q)) & 0xffu));
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int rr = 0; rr < RB; ++rr) {
            const int r = RB * h + rr;
            const float wq[4] = {sw[r].x, sw[r].y, sw[r].z, sw[r].w};
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                T4[0][r] = fmaf(wq[q], a[rr][q].x, T4[0][r]); T4[1][r] = fmaf(wq[q], a[rr][q].y, T4[1][r]);
                T4[2][r] = fmaf(wq[q], a[rr][q].z, T4[2][r]); T4[3][r] = fmaf(wq[q], a[rr][q].w, T4[3][r]);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    }
}

// SPARSE (round 4): skinning with the <= 4 non-zero weights of a vertex (nemo_ctx::d_Wsp_*) on the VALU -- per (vertex,
// sample) 4 x 12 FMAs against float4 rows of the joints' transforms in LDS -- instead of the dense 24-joint MFMA product:
// 144 of the tile's 552 fp32 MFMAs (26 % of its matrix-pipe cycles) are gone, and the VALU work (384 FMAs + 96
// ds_read_b128 per lane and tile) runs under the other resident wave's MFMAs.  Same sums without the zero terms.
template <int MODE, bool SPARSE = false>
__global__ __launch_bounds__(256, 2) void mesh_v2v_fused_kernel(
    long N, long NV, const float* __restrict__ PF2, long ldpf, const float* __restrict__ A2,
    const float* __restrict__ P, long ldP, const float* __restrict__ vs, const float* __restrict__ W,
    int G, int cpg, int RA, int CA, int nB, int vec_stage, float* __restrict__ loss_sum, float* __restrict__ dVPt, long ldn,
    float* __restrict__ dA, float* __restrict__ parts, int* __restrict__ tickets, float* __restrict__ loss_parts,
    int* __restrict__ grid_ticket, unsigned short* __restrict__ dVPb, long ldk,
    const unsigned short* __restrict__ Wsk, const unsigned short* __restrict__ Wadj,
    const float* __restrict__ Wsp_w, const unsigned int* __restrict__ Wsp_j, float pscale, long hplane) {
    // MODE 6 (round 6): MODE 5 whose two skinnings run as split-precision MFMAs too -- T = W A with W as two fp16 pieces of 2^14 W and the
    // transforms as two fp16 pieces of 2^12 A (three piece products: fp32-equivalent), the dense 24-joint product of lbs.py:236-241 on the
    // 16-bit pipe instead of the <= 4 non-zero weights on the VALU (768 FMAs per tile and lane); SKH marks it inside the SPLIT paths
    constexpr bool BF16 = MODE >= 1 && MODE <= 3, SKH = MODE == 6, SPLIT = MODE == 2 || SKH, ADJS = MODE == 2 || MODE == 3;
    constexpr float A_SCALE = SKH ? 4096.f : 1.f, T_UNSCALE = SKH ? 1.f / (16384.f * 4096.f) : 1.f;
    // MODE 4 (round 5, "f32_split"): fp32 arithmetic everywhere EXCEPT that the pose blend's products run on the bf16 pipe with
    // both operands carried as THREE bf16 pieces (8 + 8 + 8 significant bits = the fp32 value): x = x0 + x1 + x2, and
    // P pf = P0 pf0 + P0 pf1 + P1 pf0 + P0 pf2 + P1 pf1 + P2 pf0 (+ terms below 2^-24 of |P| |pf|), every piece product exact in
    // fp32, fp32 accumulation: 252 MFMAs of 16 cycles per tile instead of 312 of 32.  P then points at
    // nemo_ctx::d_posedirs_sp3, pose features are split when they are staged.
    // MODE 5 (round 5, last third; the form `mesh_blend = 'f32_split'` runs): the same with TWO fp16 pieces per operand --
    // x0 = fp16(s x), x1 = fp16(s x - x0) with a power-of-two scale s that keeps the pieces in fp16's normal range: 11 + 11
    // significant bits + the sign of the remainder = 23 of the fp32 value's 24 bits in the worst case (|s x - x0 - x1| <= 2^-23 |s x|,
    // one fp32 ulp; tests/test_split_precision.py).  A product keeps x0 y0 + x0 y1 + x1 y0 (the dropped x1 y1 <= 2^-22 |x y|), each
    // exact in fp32 (11 x 11 bits): 126 MFMAs of 16 cycles for the blend instead of 252, two thirds of the blend-shape bytes and of the
    // LDS.  Per term that is up to 2^-21 against the 2^-24 of an fp32 FMA; over the 207 terms of a blend the fp32 accumulation's own
    // rounding dominates either way -- error against float64: fp32 product 5.1e-7, this 5.0e-7, three bf16 pieces 3.2e-7 of the result's
    // scale (same test), and the kernel's outputs equal the fp32-MFMA kernel's error (test_v2v_fused_split_is_fp32_equivalent).
    constexpr bool SP3 = MODE == 4 || MODE == 5 || MODE == 6, SPH = MODE == 5 || MODE == 6, B16 = BF16 || SP3;
    constexpr int NP = SPH ? 2 : 3;                          // pieces per operand
    using e16 = std::conditional_t<SPH, _Float16, __bf16>;   // their element type
    auto mfma_p = [](const u32x4m& a, const u32x4m& b, const f32x4& c) {     // one piece product on the matrix pipe
        if constexpr (SPH) return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(mf16x8, a), __builtin_bit_cast(mf16x8, b), c, 0, 0, 0);
        else return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(mbf16x8, a), __builtin_bit_cast(mbf16x8, b), c, 0, 0, 0);
    };
    constexpr float PF_SCALE = SPH ? 4096.f : 1.f;           // pose features (|pf| <= 2) and vp (|vp| < 4 m) are staged x 2^12
    const float blend_scale = SPH ? pscale * PF_SCALE : 1.f; // the blend accumulates scale(P) * scale(pf) * (P pf)
    // ... and the vertex->joint adjoint dA = W^T dT likewise (ADJ3): W in three pieces as A images [W0 | W0], [W1 | W1], [W0 | W2]
    // over the two 16-vertex slots of a K = 32 instruction (Wadj then points at nemo_ctx::d_Wadj3), dT in three pieces as B
    // operands [T0 ; T1] and [T2 ; T0]: three MFMAs of 16 cycles per (entry, joint tile) instead of eight fp32 ones of 32.
    // dT_e = gs * vp_d with gs in {-1, 0, +1}: its pieces are the pieces of vp -- formed ONCE per tile -- with the sign bit
    // flipped / zeroed per row.
#ifndef MESH_ADJ3
#define MESH_ADJ3 1
#endif
    constexpr bool ADJ3 = SP3 && MESH_ADJ3 != 0;
    // Placement (measured, 8 x 300): the adjoint MFMAs of all three rows BEHIND the row loop and the dvp store, the three W images
    // requested there too: no spilled register, 374 us per launch against 386 with the fp32 adjoint (dense weights: 416 against 462).
    // Inside the row loop, or with the W images requested earlier, hipcc spills 33 - 231 registers of the sparse instantiation
    // (474 - 635 us per launch; profiles/r05_experiments.md section 5).
    static_assert(!(SPARSE && SPLIT), "sparse skinning replaces the split-precision skinning");
    // All constants are zero-padded by nemo_ctx_create (P: 224 rows x 3*NVp columns, W / v_shaped: NVp
    // vertices) and dVPt has 3*NVp rows x ldn >= 16*groups columns, so no load or store below needs a
    // predicate: padded vertices / samples produce v_rec == v_orig == 0, i.e. zero loss and gradient.
    // The ONLY LDS object of the kernel: dynamic LDS starts where the static objects end, whatever alignment it declares,
    // and the three small ones this kernel used to have (72 bytes) left every ds_read_b128 below 8-byte aligned --
    // 64 LDS cycles per instruction instead of 4 (SQ_LDS_UNALIGNED_STALL = 90 % of the LDS-active cycles).  They live
    // in the last 32 floats of the dynamic block now (MF_TAIL).
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* pfL = lds;                                   // [2][16][MF_PFS] floats  (BF16: [2][16][MF_PFB] bf16)
    __bf16* pfB = reinterpret_cast<__bf16*>(lds);
    float* AL = lds + (SP3 ? NP * 2 * 16 * MF_PFB / 2 : BF16 ? 2 * 16 * MF_PFB / 2 : 2 * 16 * MF_PFS);   // [2][16][MF_AS], entry (e*24 + j)
                                                        // (SP3: pose features as [piece NP][body 2][16][MF_PFB] bf16 / fp16)
    __bf16* ALb = reinterpret_cast<__bf16*>(AL);        // MODE 2: [body 2][piece 2][16][MF_AB] bf16, entry (e*32 + j)
    float* red = AL + (SPLIT ? 2 * 2 * 16 * MF_AB / 2 : 2 * 16 * (SPARSE ? MF_ASP : MF_AS));   // MF_TAIL: 16 floats + the two flags
    int& ticket_old = *reinterpret_cast<int*>(red + 16);
    int& grid_last = *reinterpret_cast<int*>(red + 17);
    const long ntiles = (NV + 15) / 16;
    // Work = (sample group, chunk of 4 vertex tiles) units, G groups x cpg chunks.  Two classes of blocks, all
    // co-resident (<= 512 = 2 per CU), planned on the host (mesh_plan) so that every block gets the same
    // number of chunks:
    //   A: bid < G*RA      one segment: group bid % G, the r-th of RA equal ranges of the chunks [0, CA)
    //   B: bid >= G*RA     the left-over range [CA, cpg) of up to k consecutive groups, one segment per group
    // Blocks of one range sweep the same vertex tiles in lock step, so a blend-shape tile is fetched into
    // an XCD's L2 once per range and hit by the other blocks (an earlier 1-D split at arbitrary chunk
    // offsets had every block on its own tile position: 10x the L2 fill traffic).  A group's partial dA
    // sums come from its RA range blocks + its left-over segment.
    // From 128 sample groups on, workgroups b and b + 256 -- the two of a CU -- take ADJACENT positions of the (range, group) order,
    // i.e. the same vertex range: their blend-shape loads meet in the CU's L1 (round 5; mesh launch at 8 / 16 / 40 x 300:
    // 369 -> 360, 760 -> 737, 854 -> 821 us; with few groups and many ranges it costs instead -- 1 x 300: 90 -> 98 us -- and the
    // plan's pairing of long with short ranges, below, is the better use of the two slots).
    const int nbk = (int)gridDim.x, mpair = nbk > 256 ? nbk - 256 : 0, bx = (int)blockIdx.x;
    const int bid = G < 128 ? bx : (bx < 256 ? (bx < mpair ? 2 * bx : 2 * mpair + (bx - mpair)) : 2 * (bx - 256) + 1);
    const int GA = G * RA;
    int g_lo, nseg, k_beg, k_end, slot;
    if (bid < GA) {
        const int r = bid / G;
        g_lo = bid - r * G; nseg = 1; slot = r;
        // CA % RA ranges are one chunk longer; they come FIRST, so that (blocks b and b + 256 sharing a CU) a long range
        // is never paired with another long one
        const int q = CA / RA, rem = CA - q * RA;
        k_beg = r * q + min(r, rem); k_end = k_beg + q + (r < rem ? 1 : 0);
    } else {
        const int j = bid - GA;
        g_lo = (int)(((long)G * j) / nB); nseg = (int)(((long)G * (j + 1)) / nB) - g_lo; slot = RA;
        k_beg = CA; k_end = cpg;
    }
    const int nr = RA + (CA < cpg ? 1 : 0), maxc = RA + 1;
    float lsum = 0.f;
#pragma unroll 1
    for (int seg = 0; seg < nseg; ++seg) {
    // lane-derived offsets are re-derived per segment (the asm keeps hipcc from hoisting them -- and the
    // address arithmetic that hangs off them -- out of the loop, which cost > 100 spilled VGPRs)
    int tid = threadIdx.x;
    asm volatile("" : "+v"(tid));
    const int lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave-uniform (SGPR): tile indices, offsets
    const int l15 = lane & 15, g = lane >> 4;
    const int grp = g_lo + seg;
    const long s0 = (long)grp * 16;
    const long t_beg = 4 * (long)k_beg;
    const long t_end = min(ntiles, 4 * (long)k_end);
    if (seg) __syncthreads();                             // LDS of the previous segment fully consumed

    // ---- stage the sample group: pf rows (224 incl. the zero pad) and transforms re-ordered to [e][j].
    // Vector path (16-byte aligned rows): a fixed number of independent, branch-free dwordx4 loads per
    // thread, all in flight together -- a block that owns only a few vertex tiles (small N: many vertex
    // ranges per sample group) would otherwise spend more time in a scalar staging loop than in MFMAs.
    if (vec_stage) {
        float4 vpf[7], va[9];
#pragma unroll
        for (int it = 0; it < 7; ++it) {                            // 2 x 16 x 52 float4 of pose features
            const int idx = min(tid + 256 * it, 2 * 16 * 52 - 1);
            const int set = idx / 832, n = (idx / 52) % 16, q = idx % 52;
            const long sc = min(s0 + n, N - 1);
            vpf[it] = *reinterpret_cast<const float4*>(PF2 + (set * N + sc) * ldpf + 4 * q);
        }
#pragma unroll
        for (int it = 0; it < 9; ++it) {                            // 2 x 16 x 72 float4 of transforms
            const int idx = tid + 256 * it;
            const int set = idx / 1152, n = (idx / 72) % 16, q = idx % 72;
            const long sc = min(s0 + n, N - 1);
            va[it] = *reinterpret_cast<const float4*>(A2 + (set * N + sc) * 288 + 4 * q);
        }
#pragma unroll
        for (int it = 0; it < 7; ++it) {
            const int idx = min(tid + 256 * it, 2 * 16 * 52 - 1);
            const int set = idx / 832, n = (idx / 52) % 16, q = idx % 52;
            const bool live = s0 + n < N;
            float4 v = vpf[it];
            if (!live) v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (q == 51) v.w = 0.f;                                 // column 207 is padding
            if (SP3) {
                typedef e16 e4 __attribute__((ext_vector_type(4)));
                float x[4] = {v.x * PF_SCALE, v.y * PF_SCALE, v.z * PF_SCALE, v.w * PF_SCALE};
#pragma unroll
                for (int pc = 0; pc < NP; ++pc) {
                    e4 h;
#pragma unroll
                    for (int i = 0; i < 4; ++i) { h[i] = (e16)x[i]; x[i] -= (float)h[i]; }
                    *reinterpret_cast<e4*>(pfB + ((pc * 2 + set) * 16 + n) * MF_PFB + 4 * q) = h;
                }
            } else if (BF16) {
                typedef __bf16 bf4 __attribute__((ext_vector_type(4)));
                const bf4 h = {(__bf16)v.x, (__bf16)v.y, (__bf16)v.z, (__bf16)v.w};
                *reinterpret_cast<bf4*>(pfB + (set * 16 + n) * MF_PFB + 4 * q) = h;          // 8-byte store
            } else {
                float* d = pfL + (set * 16 + n) * MF_PFS + 4 * q;       // 8-byte aligned (MF_PFS even)
                *reinterpret_cast<float2*>(d) = make_float2(v.x, v.y);
                *reinterpret_cast<float2*>(d + 2) = make_float2(v.z, v.w);
            }
        }
        for (int idx = tid; idx < (SP3 ? NP : 1) * 2 * 16 * 16; idx += 256) {       // rows 208..223 of the k padding
            if (B16) pfB[(idx / 16) * MF_PFB + 208 + idx % 16] = (__bf16)0.f;
            else pfL[(idx / 16) * MF_PFS + 208 + idx % 16] = 0.f;
        }
#pragma unroll
        for (int it = 0; it < 9; ++it) {
            const int idx = tid + 256 * it;
            const int set = idx / 1152, n = (idx / 72) % 16, q = idx % 72;
            const int j = q / 3, e = 4 * (q % 3);                   // a float4 never straddles a joint
            const bool live = s0 + n < N;
            if constexpr (SPARSE) {
                float4 v = va[it];
                if (!live) v = make_float4(0.f, 0.f, 0.f, 0.f);
                *reinterpret_cast<float4*>(AL + ((set * 72 + q) * 16 + n) * 4) = v;
            } else if constexpr (SPLIT) {
                const float x[4] = {va[it].x, va[it].y, va[it].z, va[it].w};
                __bf16* dh = ALb + ((set * 2 + 0) * 16 + n) * MF_AB + e * 32 + j;
                __bf16* dl = ALb + ((set * 2 + 1) * 16 + n) * MF_AB + e * 32 + j;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float xv = (live ? x[i] : 0.f) * A_SCALE;
                    if constexpr (SKH) {
                        const _Float16 h = (_Float16)xv;
                        dh[32 * i] = __builtin_bit_cast(__bf16, h); dl[32 * i] = __builtin_bit_cast(__bf16, (_Float16)(xv - (float)h));
                    } else {
                        const __bf16 h = (__bf16)xv;
                        dh[32 * i] = h; dl[32 * i] = (__bf16)(xv - (float)h);
                    }
                }
            } else {
            float* d = AL + (set * 16 + n) * MF_AS + e * 24 + j;
            d[0] = live ? va[it].x : 0.f; d[24] = live ? va[it].y : 0.f;
            d[48] = live ? va[it].z : 0.f; d[72] = live ? va[it].w : 0.f;
            }
        }
    } else {
        for (int idx = tid; idx < 2 * 16 * 224; idx += 256) {
            const int set = idx / (16 * 224), n = (idx / 224) % 16, p = idx % 224;
            const long s = s0 + n;
            const float val = (s < N && p < 207) ? PF2[(set * N + s) * ldpf + p] : 0.f;
            if (SP3) {
                float x = val * PF_SCALE;
#pragma unroll
                for (int pc = 0; pc < NP; ++pc) {
                    const e16 h = (e16)x;
                    pfB[((pc * 2 + set) * 16 + n) * MF_PFB + p] = __builtin_bit_cast(__bf16, h);
                    x -= (float)h;
                }
            } else if (BF16) pfB[(set * 16 + n) * MF_PFB + p] = (__bf16)val;
            else pfL[(set * 16 + n) * MF_PFS + p] = val;
        }
        for (int idx = tid; idx < 2 * 16 * 288; idx += 256) {
            const int set = idx / (16 * 288), n = (idx / 288) % 16, je = idx % 288;
            const int j = je / 12, e = je % 12;
            const long s = s0 + n;
            const float xv = (s < N) ? A2[(set * N + s) * 288 + je] : 0.f;
            if constexpr (SPARSE) {
                AL[((set * 72 + je / 4) * 16 + n) * 4 + je % 4] = xv;
            } else if constexpr (SKH) {
                const float xs = xv * A_SCALE;
                const _Float16 h = (_Float16)xs;
                ALb[((set * 2 + 0) * 16 + n) * MF_AB + e * 32 + j] = __builtin_bit_cast(__bf16, h);
                ALb[((set * 2 + 1) * 16 + n) * MF_AB + e * 32 + j] = __builtin_bit_cast(__bf16, (_Float16)(xs - (float)h));
            } else if constexpr (SPLIT) {
                const __bf16 h = (__bf16)xv;
                ALb[((set * 2 + 0) * 16 + n) * MF_AB + e * 32 + j] = h;
                ALb[((set * 2 + 1) * 16 + n) * MF_AB + e * 32 + j] = (__bf16)(xv - (float)h);
            } else {
                AL[(set * 16 + n) * MF_AS + e * 24 + j] = xv;
            }
        }
    }
    if constexpr (SPLIT) {
        // joints 24..31 of every (body, piece, sample, entry) row: the zero tail of the K = 32 instruction
        for (int idx = tid; idx < 2 * 2 * 16 * 12; idx += 256) {
            const int row = idx / 12, e = idx % 12;
            *reinterpret_cast<uint4*>(ALb + row * MF_AB + e * 32 + 24) = make_uint4(0u, 0u, 0u, 0u);
        }
    }
    __syncthreads();

    f32x4 accdA[12][2];
#pragma unroll
    for (int e = 0; e < 12; ++e)
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) accdA[e][t][r] = 0.f;
    const float* pf0 = pfL + (0 * 16 + l15) * MF_PFS + g;      // orig body, this lane's sample column
    const float* pf1 = pfL + (1 * 16 + l15) * MF_PFS + g;      // reconstruction
    const float* A0 = SPARSE ? AL + 4 * l15 : AL + (0 * 16 + l15) * MF_AS + g;
    const float* A1 = SPARSE ? AL + 72 * 64 + 4 * l15 : AL + (1 * 16 + l15) * MF_AS + g;
    const __bf16* Ab0 = ALb + ((0 * 2 + 0) * 16 + l15) * MF_AB + 8 * g;   // MODE 2: hi piece; the lo piece is 16 rows on
    const __bf16* Ab1 = ALb + ((1 * 2 + 0) * 16 + l15) * MF_AB + 8 * g;

    // Blend-shape A-operands, eight k-steps in flight, carried ACROSS vertex tiles: the end of a tile
    // requests the first eight k-steps of the wave's next tile, so a tile never starts with an empty
    // ring (one exposed L2 round trip per tile otherwise).  Addresses are a wave-uniform row base plus one
    // 32-bit lane offset; a scheduling barrier per k-step keeps hipcc from collapsing the ring to a
    // single load in flight (it otherwise moves each load to just before its use to save registers).
    // Buffer loads: one descriptor for the blend-shape matrix, a scalar byte offset for (tile, k-step) and
    // ONE 32-bit lane offset -- no 64-bit vector address arithmetic and no address registers in the ring.
    // (BF16: P = bf16 blend shapes [tile][S][component][g][vertex][8 k], ldP = NVp; one dwordx4 per lane, component and MFMA:
    //  1 KB of consecutive memory per instruction)
    const __amdgpu_buffer_rsrc_t Prs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(P), 0, SP3 ? (int)(28 * ldP * 48 * NP) : BF16 ? (int)(28 * ldP * 48) : (int)(224 * ldP * 4), 0x00020000);
    const int loff = B16 ? lane * 16 : (g * (int)ldP + l15 * 3) * 4;                    // lane part of the address (bytes)
    const int kstride = SP3 ? 3 * NP * 1024 : BF16 ? 3 * 1024 : 4 * (int)ldP * 4;            // bytes between consecutive k-steps
    constexpr int TILE_B = SP3 ? 7 * 3 * NP * 1024 : BF16 ? 7 * 3 * 1024 : 192;              // bytes between consecutive vertex tiles
    // (SP3: P = [tile][S][component][piece][g][vertex][8 k] bf16, 1 KB per (S, component, piece))
    u32x3 pa[8];                                                 // fp32: eight k-steps (of 4) in flight
    constexpr int PQD = MODE == 3 ? MESH_PQD : 2;                      // bf16: PQD k-steps (of 32) x 3 components in flight
    u32x4m pq[PQD][3];
    u32x4m ps[3][3];                                             // SP3: ONE k-step, [component][piece 1, 2], refilled piece by piece
    u32x4m ps0[2][3];                                            // SP3: piece 0 (half of a k-step's products), double-buffered [k-step & 1][component]
    auto sp3_request = [&](const int soff, const int buf0) {     // all 3 NP pieces of one k-step (scalar byte offset soff)
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            ps0[buf0][c] = __builtin_amdgcn_raw_buffer_load_b128(Prs, loff, soff + (NP * c) * 1024, 0);
#pragma unroll
            for (int pc = 1; pc < NP; ++pc) ps[c][pc] = __builtin_amdgcn_raw_buffer_load_b128(Prs, loff, soff + (NP * c + pc) * 1024, 0);
        }
    };
    {
        const int tf = (int)min(t_beg + wid, ntiles - 1);        // (a wave without tiles loads a valid one)
        if (SP3) {
            sp3_request(tf * TILE_B, 0);
        } else if (BF16) {
#pragma unroll
            for (int u = 0; u < PQD; ++u)
#pragma unroll
                for (int c = 0; c < 3; ++c)
                    pq[u][c] = __builtin_amdgcn_raw_buffer_load_b128(Prs, loff, tf * TILE_B + u * kstride + 1024 * c, 0);
        } else {
#pragma unroll
            for (int u = 0; u < 8; ++u) pa[u] = __builtin_amdgcn_raw_buffer_load_b96(Prs, loff, tf * 192 + u * kstride, 0);
        }
    }
    /*prof:init*/
    for (long t = t_beg + wid; t < t_end; t += 4) {
        /*prof:c0*/
        const long v0 = t * 16;
        // weights as A-operands: forward (rows = vertices, k = joints) and adjoint (rows = joints, k = vertices)
        float wf[6], wa[4][2];
        mbf16x8 wsk[SKH ? 3 : 2], wad[2][2];                     // split precision: [piece], [joint tile][piece]
        u32x4m wad3[2][NP];                                      // ADJ3: [joint tile][A image]
        float4 sw[4];                                            // sparse: this lane's 4 vertices x <= 4 (weight, joint)
        unsigned int sj[4];
        if constexpr (SPARSE) {
        } else if constexpr (SPLIT) {
            const long NVp16 = ((NV + 15) / 16) * 16;
#pragma unroll
            for (int pc = 0; pc < (SKH ? 3 : 2); ++pc)
                wsk[pc] = *reinterpret_cast<const mbf16x8*>(Wsk + ((long)pc * NVp16 + v0 + l15) * 32 + 8 * g);
        } else {
            const float* Wf = W + (v0 + l15) * 24 + g;
#pragma unroll
            for (int kk = 0; kk < 6; ++kk) wf[kk] = Wf[4 * kk];
        }
        auto load_adjoint_weights = [&]() {
        if constexpr (ADJ3) {
#pragma unroll
            for (int jt = 0; jt < 2; ++jt)
#pragma unroll
                for (int q = 0; q < NP; ++q)
                    wad3[jt][q] = *reinterpret_cast<const u32x4m*>(Wadj + (((t * 2 + jt) * NP + q) * 64 + lane) * 8);
        } else if constexpr (ADJS) {
#pragma unroll
            for (int jt = 0; jt < 2; ++jt)
#pragma unroll
                for (int pc = 0; pc < 2; ++pc)
                    wad[jt][pc] = *reinterpret_cast<const mbf16x8*>(Wadj + (((t * 2 + jt) * 2 + pc) * 64 + lane) * 8);
        } else {
            const float* Wa = W + (v0 + 4 * g) * 24 + l15;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                wa[r][0] = Wa[r * 24];
                wa[r][1] = l15 < 8 ? Wa[r * 24 + 16] : 0.f;
            }
        }
        };
        // (sparse bf16: requested after the reconstruction body -- first used a skinning row + the sign pass later -- so that
        //  their 16 registers are not live across the blend and the first skinning phase: 18 spilled registers otherwise.
        //  The fp32 form keeps them at the top of the tile: moved, hipcc spills 270.)
        if constexpr (!(SPARSE && ADJS) && !ADJ3) load_adjoint_weights();
        // ---- pose blend of both bodies: 52 k-steps x 3 components, A-operand P[p][3v+c] from L2
        f32x4 vp[2][3];
        const float* vsl = vs + (v0 + 4 * g) * 3;
#pragma unroll
        for (int c = 0; c < 3; ++c)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float b = vsl[r * 3 + c];
                vp[0][c][r] = b; vp[1][c][r] = b;
            }
        const int pt = (int)t * TILE_B;                          // uniform byte offsets: this wave's vertex tile
        const int ptn = (int)min(t + 4, ntiles - 1) * TILE_B;    // and its next one (clamped: harmless re-read)
        if constexpr (SP3) {
            // 7 k-steps of 32 x NP (NP + 1) / 2 piece products x 3 components x 2 bodies = 252 (three bf16 pieces) / 126 (two fp16
            // pieces) MFMAs.  Product order per k-step: those with P's piece 0, then its LAST piece, ..., piece 1 (why: at the loop).
            // Six accumulators in rotation: no MFMA waits for its predecessor.
            // (B operands -- the pose-feature pieces, LDS -- one k-step ahead: requested behind the first product group of the
            //  previous k-step)
            if constexpr (SPH) {                                 // (the accumulators carry scale(P) scale(pf) x the sums)
#pragma unroll
                for (int c = 0; c < 3; ++c) { vp[0][c] *= blend_scale; vp[1][c] *= blend_scale; }
            }
            u32x4m b[2][2][NP];                                  // [buffer][body][piece]
            auto load_b = [&](const int buf, const int S) {
#pragma unroll
                for (int pc = 0; pc < NP; ++pc)
#pragma unroll
                    for (int bd = 0; bd < 2; ++bd)
                        b[buf][bd][pc] = *reinterpret_cast<const u32x4m*>(pfB + ((pc * 2 + bd) * 16 + l15) * MF_PFB + 8 * g + 32 * S);
            };
            load_b(0, 0);
#pragma unroll
            for (int S = 0; S < 7; ++S) {
                // piece 0 carries half (two thirds) of a k-step's products: re-requested behind them it would have 18 (6) MFMAs until
                // its next use, less than an L2 round trip -- so it has two buffers and is requested a whole k-step ahead; the other
                // pieces are re-requested behind their products (three pieces: 2 then 1, 30 and 24 MFMAs ahead; two: 12 ahead)
                if (S + 1 < 7) {
#pragma unroll
                    for (int c = 0; c < 3; ++c)
                        ps0[(S + 1) & 1][c] = __builtin_amdgcn_raw_buffer_load_b128(Prs, loff, pt + (S + 1) * kstride + (NP * c) * 1024, 0);
                }
#pragma unroll
                for (int gi = 0; gi < NP; ++gi) {
                    const int pa = gi == 0 ? 0 : NP - gi;
#pragma unroll
                    for (int pb = NP - 1 - pa; pb >= 0; --pb)
#pragma unroll
                        for (int c = 0; c < 3; ++c) {
                            const u32x4m a = pa == 0 ? ps0[S & 1][c] : ps[c][pa];
                            vp[0][c] = mfma_p(a, b[S & 1][0][pb], vp[0][c]);
                            vp[1][c] = mfma_p(a, b[S & 1][1][pb], vp[1][c]);
                        }
                    if (S + 1 < 7) {
                        if (pa != 0) {
#pragma unroll
                            for (int c = 0; c < 3; ++c)
                                ps[c][pa] = __builtin_amdgcn_raw_buffer_load_b128(Prs, loff, pt + (S + 1) * kstride + (NP * c + pa) * 1024, 0);
                        } else {
                            load_b((S + 1) & 1, S + 1);
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            if constexpr (SPH) {
                const float inv = 1.f / blend_scale;             // (a power of two: exact)
#pragma unroll
                for (int c = 0; c < 3; ++c) { vp[0][c] *= inv; vp[1][c] *= inv; }
            }
        } else if constexpr (BF16) {
            // 7 k-steps of 32 (207 blend shapes + zero rows) x 3 components x 2 bodies = 42 MFMAs
            const __bf16* pb0 = pfB + (0 * 16 + l15) * MF_PFB + 8 * g;
            const __bf16* pb1 = pfB + (1 * 16 + l15) * MF_PFB + 8 * g;
#pragma unroll
            for (int S = 0; S < 7; ++S) {
                const mbf16x8 b0 = *reinterpret_cast<const mbf16x8*>(pb0 + 32 * S);
                const mbf16x8 b1 = *reinterpret_cast<const mbf16x8*>(pb1 + 32 * S);
                mbf16x8 a[3];
#pragma unroll
                for (int c = 0; c < 3; ++c) a[c] = __builtin_bit_cast(mbf16x8, pq[S % PQD][c]);
                if (S + PQD < 7) {
#pragma unroll
                    for (int c = 0; c < 3; ++c)
                        pq[S % PQD][c] = __builtin_amdgcn_raw_buffer_load_b128(Prs, loff, pt + (S + PQD) * kstride + 1024 * c, 0);
                }
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    vp[0][c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[c], b0, vp[0][c], 0, 0, 0);
                    vp[1][c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[c], b1, vp[1][c], 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        } else {
        // B-operands (this lane's pose features, LDS) two k-steps ahead of their use
        float pb[2][2];
#pragma unroll
        for (int u = 0; u < 2; ++u) { pb[u][0] = pf0[4 * u]; pb[u][1] = pf1[4 * u]; }
#pragma unroll
        for (int kk0 = 0; kk0 < 56; kk0 += 8) {                  // 52 k-steps (207 blend shapes + one zero row):
#pragma unroll                                                   // fully unrolled, the last group is half a group
            for (int u = 0; u < 8; ++u) {
                if (kk0 + u >= 52) continue;
                const float a0 = __uint_as_float(pa[u][0]), a1 = __uint_as_float(pa[u][1]),
                            a2 = __uint_as_float(pa[u][2]);
                const float b0 = pb[u & 1][0], b1 = pb[u & 1][1];
                if (kk0 + u + 2 < 52) { pb[u & 1][0] = pf0[4 * (kk0 + u + 2)]; pb[u & 1][1] = pf1[4 * (kk0 + u + 2)]; }
                if (kk0 + 8 + u < 52)                                               // k-step kk + 8
                    pa[u] = __builtin_amdgcn_raw_buffer_load_b96(Prs, loff, pt + (kk0 + 8 + u) * kstride, 0);
                vp[0][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, b0, vp[0][0], 0, 0, 0);
                vp[1][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, b1, vp[1][0], 0, 0, 0);
                vp[0][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, b0, vp[0][1], 0, 0, 0);
                vp[1][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, b1, vp[1][1], 0, 0, 0);
                vp[0][2] = __builtin_amdgcn_mfma_f32_16x16x4f32(a2, b0, vp[0][2], 0, 0, 0);
                vp[1][2] = __builtin_amdgcn_mfma_f32_16x16x4f32(a2, b1, vp[1][2], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        }
        /*prof:c1*/
        if constexpr (SPARSE) {
            // (loaded HERE, after the blend: 20 registers the blend loop's operand ring needs; requesting them four
            //  k-steps before the end of the blend instead measured the same, 501 against 494 - 502 us)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                sw[r] = *reinterpret_cast<const float4*>(Wsp_w + (v0 + 4 * g + r) * 4);
                sj[r] = Wsp_j[v0 + 4 * g + r];
            }
        }
        // ---- reconstruction body, one output row c (4 transform entries) at a time
        float vrec[3][4];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            f32x4 T4[4];
#pragma unroll
            for (int d = 0; d < 4; ++d)
#pragma unroll
                for (int r = 0; r < 4; ++r) T4[d][r] = 0.f;
            if constexpr (SPARSE) {
                sparse_rows<2>(A1 + 64 * c, sw, sj, T4);
            } else if constexpr (SPLIT) {
                mbf16x8 bh[4], bl[4];
#pragma unroll
                for (int d = 0; d < 4; ++d) {
                    bh[d] = *reinterpret_cast<const mbf16x8*>(Ab1 + (4 * c + d) * 32);
                    bl[d] = *reinterpret_cast<const mbf16x8*>(Ab1 + 16 * MF_AB + (4 * c + d) * 32);
                }
                auto mf = [](const mbf16x8& a, const mbf16x8& b, const f32x4& acc) {
                    if constexpr (SKH) return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(mf16x8, a), __builtin_bit_cast(mf16x8, b), acc, 0, 0, 0);
                    else return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc, 0, 0, 0);
                };
                if constexpr (SKH) {         // W exact (three pieces), A to 2^-23; W1 A1 (<= 2^-24 of the product) is dropped
#pragma unroll
                    for (int d = 0; d < 4; ++d) T4[d] = mf(wsk[2], bh[d], T4[d]);
                }
#pragma unroll
                for (int d = 0; d < 4; ++d) T4[d] = mf(wsk[0], bl[d], T4[d]);          // (minor products first)
#pragma unroll
                for (int d = 0; d < 4; ++d) T4[d] = mf(wsk[1], bh[d], T4[d]);
#pragma unroll
                for (int d = 0; d < 4; ++d) T4[d] = mf(wsk[0], bh[d], T4[d]);
                // (SKH: T4 stays 2^26 T here -- the power of two leaves with the row's result / the sign, 12 multiplies per row instead of 32)
            } else {
#pragma unroll
            for (int kk = 0; kk < 6; ++kk)
#pragma unroll
                for (int d = 0; d < 4; ++d)       // 4 independent accumulators back to back
                    T4[d] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[kk], A1[(4 * c + d) * 24 + 4 * kk], T4[d], 0, 0, 0);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r)
                vrec[c][r] = (T4[0][r] * vp[1][0][r] + T4[1][r] * vp[1][1][r] + T4[2][r] * vp[1][2][r] + T4[3][r]) * T_UNSCALE;
            __builtin_amdgcn_sched_barrier(0);
        }
        /*prof:c2*/
        if constexpr (SPARSE && ADJS) load_adjoint_weights();
        unsigned int smz[3][2], zmz[3][2];                       // ADJ3: sign / zero masks of the three rows, [row][vertex pair]
        unsigned int vpp[NP][3][2];                              // ADJ3: [piece][coordinate d][vertex pair] of vp_orig, packed 16-bit pieces
        auto adj3_prefetch = [&]() { sp3_request(ptn, 0); };     // the wave's NEXT tile: its first k-step
        auto adj3_pieces = [&]() {
#pragma unroll
            for (int d = 0; d < 3; ++d) {
                float x[4] = {vp[0][d][0] * PF_SCALE, vp[0][d][1] * PF_SCALE, vp[0][d][2] * PF_SCALE, vp[0][d][3] * PF_SCALE};
#pragma unroll
                for (int pc = 0; pc < NP; ++pc)
#pragma unroll
                    for (int pr = 0; pr < 2; ++pr) {
                        const e16 h0 = (e16)x[2 * pr], h1 = (e16)x[2 * pr + 1];
                        vpp[pc][d][pr] = (unsigned int)__builtin_bit_cast(unsigned short, h0) |
                                         ((unsigned int)__builtin_bit_cast(unsigned short, h1) << 16);
                        x[2 * pr] -= (float)h0; x[2 * pr + 1] -= (float)h1;
                    }
            }
        };
        auto adj3_row = [&](const int c) {
#pragma unroll
            for (int dh = 0; dh < 2; ++dh) {                     // two entries (c, d) at a time: four accumulators in rotation
                u32x4m b1[2], b2[2];                             // [T0 ; T1] and (three pieces) [T2 ; T0]
#pragma unroll
                for (int dd = 0; dd < 2; ++dd) {
                    const int d = 2 * dh + dd;
                    unsigned int tk[NP][2];
#pragma unroll
                    for (int pr = 0; pr < 2; ++pr) {
                        // (d = 3: dT = gs itself, +-1.0 -- bf16 0x3f80 / fp16 PF_SCALE = 4096.0 = 0x6c00 -- with the sign flipped, no lower
                        //  pieces; vp pieces carry vp's sign: flipped when gs = -1)
                        const unsigned int sm = smz[c][pr], zm = zmz[c][pr];
                        tk[0][pr] = d < 3 ? ((vpp[0][d][pr] ^ sm) & zm) : (((SPH ? 0x6c006c00u : 0x3f803f80u) ^ sm) & zm);
#pragma unroll
                        for (int pc = 1; pc < NP; ++pc) tk[pc][pr] = d < 3 ? ((vpp[pc][d][pr] ^ sm) & zm) : 0u;
                    }
                    b1[dd] = u32x4m{tk[0][0], tk[0][1], tk[1][0], tk[1][1]};
                    b2[dd] = u32x4m{tk[NP - 1][0], tk[NP - 1][1], tk[0][0], tk[0][1]};
                }
                // A images (nemo_ctx_create): three pieces [W0 | W0], [W1 | W1], [W0 | W2] against [T0 ; T1], [T0 ; T1], [T2 ; T0];
                // two pieces [W0 | W0], [W1 | 0] against [T0 ; T1] twice
#pragma unroll
                for (int q = 0; q < NP; ++q)
#pragma unroll
                    for (int dd = 0; dd < 2; ++dd) {
                        const int e = 4 * c + 2 * dh + dd;
                        accdA[e][0] = mfma_p(wad3[0][q], q < 2 ? b1[dd] : b2[dd], accdA[e][0]);
                        accdA[e][1] = mfma_p(wad3[1][q], q < 2 ? b1[dd] : b2[dd], accdA[e][1]);
                    }
            }
        };
        // ---- original body: row c of the transform -> vertex coordinate c -> sign -> its share of dvp and
        // the four dT entries (c, 0..3), which go straight into the vertex->joint MFMA as B-operands
        float dvp[3][4];
#pragma unroll
        for (int d = 0; d < 3; ++d)
#pragma unroll
            for (int r = 0; r < 4; ++r) dvp[d][r] = 0.f;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            /*prof:q0*/
            f32x4 T4[4];
#pragma unroll
            for (int d = 0; d < 4; ++d)
#pragma unroll
                for (int r = 0; r < 4; ++r) T4[d][r] = 0.f;
            if constexpr (SPARSE) {
                sparse_rows<1>(A0 + 64 * c, sw, sj, T4);
            } else if constexpr (SPLIT) {
                mbf16x8 bh[4], bl[4];
#pragma unroll
                for (int d = 0; d < 4; ++d) {
                    bh[d] = *reinterpret_cast<const mbf16x8*>(Ab0 + (4 * c + d) * 32);
                    bl[d] = *reinterpret_cast<const mbf16x8*>(Ab0 + 16 * MF_AB + (4 * c + d) * 32);
                }
                auto mf = [](const mbf16x8& a, const mbf16x8& b, const f32x4& acc) {
                    if constexpr (SKH) return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(mf16x8, a), __builtin_bit_cast(mf16x8, b), acc, 0, 0, 0);
                    else return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc, 0, 0, 0);
                };
                if constexpr (SKH) {         // W exact (three pieces), A to 2^-23; W1 A1 (<= 2^-24 of the product) is dropped
#pragma unroll
                    for (int d = 0; d < 4; ++d) T4[d] = mf(wsk[2], bh[d], T4[d]);
                }
#pragma unroll
                for (int d = 0; d < 4; ++d) T4[d] = mf(wsk[0], bl[d], T4[d]);          // (minor products first)
#pragma unroll
                for (int d = 0; d < 4; ++d) T4[d] = mf(wsk[1], bh[d], T4[d]);
#pragma unroll
                for (int d = 0; d < 4; ++d) T4[d] = mf(wsk[0], bh[d], T4[d]);
                // (SKH: T4 stays 2^26 T here -- the power of two leaves with the row's result / the sign, 12 multiplies per row instead of 32)
            } else {
#pragma unroll
            for (int kk = 0; kk < 6; ++kk)
#pragma unroll
                for (int d = 0; d < 4; ++d)       // 4 independent accumulators back to back
                    T4[d] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[kk], A0[(4 * c + d) * 24 + 4 * kk], T4[d], 0, 0, 0);
            }
            /*prof:q1*/
            float gs[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float vo = (T4[0][r] * vp[0][0][r] + T4[1][r] * vp[0][1][r] + T4[2][r] * vp[0][2][r] + T4[3][r]) * T_UNSCALE;
                const float d = vrec[c][r] - vo;
                lsum += fabsf(d);
                gs[r] = d == 0.f ? 0.f : (d > 0.f ? -1.f : 1.f);          // d|v_rec - v_orig| / d v_orig
                const float gst = gs[r] * T_UNSCALE;
#pragma unroll
                for (int d2 = 0; d2 < 3; ++d2) dvp[d2][r] += T4[d2][r] * gst;
            }
            /*prof:q2*/
            if (c == 2 && !ADJ3) {
                // the wave's NEXT tile: first eight k-steps requested here, under the cover of the last 32
                // adjoint MFMAs and the dvp store (their registers are dead during the skinning phases,
                // where the pressure peaks -- a ring kept full across the whole tile spills)
                if (SP3) {
                    sp3_request(ptn, 0);
                } else if (BF16) {
#pragma unroll
                    for (int u = 0; u < PQD; ++u)
#pragma unroll
                        for (int c2 = 0; c2 < 3; ++c2)
                            pq[u][c2] = __builtin_amdgcn_raw_buffer_load_b128(Prs, loff, ptn + u * kstride + 1024 * c2, 0);
                } else {
#pragma unroll
                    for (int u = 0; u < 8; ++u)
                        pa[u] = __builtin_amdgcn_raw_buffer_load_b96(Prs, loff, ptn + u * kstride, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            if constexpr (ADJ3) {
                // sign (bit 15 / 31) and zero masks of the lane's two vertex pairs for this row; the adjoint MFMAs of all three
                // rows follow the loop (with them here, the three W images + the pieces of vp + the skinning weights: 33 - 45
                // spilled registers)
#pragma unroll
                for (int pr = 0; pr < 2; ++pr) {
                    const unsigned int g0 = __float_as_uint(gs[2 * pr]), g1 = __float_as_uint(gs[2 * pr + 1]);
                    smz[c][pr] = ((g0 >> 16) & 0x8000u) | (g1 & 0x80000000u);
                    zmz[c][pr] = (g0 ? 0xffffu : 0u) | (g1 ? 0xffff0000u : 0u);
                }
            } else if constexpr (ADJS) {
                // B-operand of entry e = (c, d): k = 8 g + t <-> the lane's own vertex row 4 g + t, hi pieces in t = 0..3,
                // lo pieces in t = 4..7 -- formed from the accumulator-layout values without leaving the lane
                mbf16x8 bq[4];
#pragma unroll
                for (int d = 0; d < 4; ++d)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float dT = d < 3 ? gs[r] * vp[0][d][r] : gs[r];
                        const __bf16 h = (__bf16)dT;
                        bq[d][r] = h; bq[d][4 + r] = (__bf16)(dT - (float)h);
                    }
#pragma unroll
                for (int pc = 0; pc < 2; ++pc)
#pragma unroll
                    for (int d = 0; d < 4; ++d) {
                        const int e = 4 * c + d;
                        accdA[e][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wad[0][pc], bq[d], accdA[e][0], 0, 0, 0);
                        accdA[e][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wad[1][pc], bq[d], accdA[e][1], 0, 0, 0);
                    }
            } else {
#pragma unroll
            for (int d = 0; d < 4; ++d) {
                const int e = 4 * c + d;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float dT = d < 3 ? gs[r] * vp[0][d][r] : gs[r];
                    accdA[e][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[r][0], dT, accdA[e][0], 0, 0, 0);
                    accdA[e][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[r][1], dT, accdA[e][1], 0, 0, 0);
                }
            }
            }
            __builtin_amdgcn_sched_barrier(0);
            /*prof:q2e*/
        }
        /*prof:q3*/
        if (SPH && dVPb) {
            // split-precision adjoint (nemo_gemm_f16x2mem_adj): d vp as TWO fp16 pieces of 2^12 d vp, NOT transposed -- plane 0 at
            // dVPb, plane 1 hplane elements behind it; row = sample, the lane's 4 vertices x 3 coordinates are 12 consecutive k
            unsigned short* dst0 = dVPb + (s0 + l15) * ldk + (v0 + 4 * g) * 3;
            const long k0 = (long)(v0 + 4 * g) * 3;          // (hplane < 0: the xp layout, see the stores below)
            unsigned short h0[12], h1[12];
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int d = 0; d < 3; ++d) {
                    const float x = dvp[d][r] * PF_SCALE;
                    const _Float16 a = (_Float16)x, b = (_Float16)(x - (float)a);
                    h0[r * 3 + d] = __builtin_bit_cast(unsigned short, a);
                    h1[r * 3 + d] = __builtin_bit_cast(unsigned short, b);
                }
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                uint2 p0, p1;
                p0.x = (unsigned)h0[4 * q] | ((unsigned)h0[4 * q + 1] << 16); p0.y = (unsigned)h0[4 * q + 2] | ((unsigned)h0[4 * q + 3] << 16);
                p1.x = (unsigned)h1[4 * q] | ((unsigned)h1[4 * q + 1] << 16); p1.y = (unsigned)h1[4 * q + 2] | ((unsigned)h1[4 * q + 3] << 16);
                if (hplane < 0) {
                    // xp matrix (csrc/gemm_xp.h fmt 2): the two pieces of 32 consecutive k are 128 consecutive bytes -- a group of four k
                    // (k0 + 4 q: a multiple of four) never straddles a k-block
                    const long kq = k0 + 4 * q;
                    unsigned short* dx = dVPb + (s0 + l15) * ldk + (kq >> 5) * 64 + (kq & 31);
                    *reinterpret_cast<uint2*>(dx) = p0;
                    *reinterpret_cast<uint2*>(dx + 32) = p1;
                } else {
                    reinterpret_cast<uint2*>(dst0)[q] = p0;
                    reinterpret_cast<uint2*>(dst0 + hplane)[q] = p1;
                }
            }
        } else if (BF16 && dVPb) {
            // bf16-in-memory chain: d vp as bf16, NOT transposed -- row = sample, the lane's 4 vertices x 3 coordinates are
            // 12 consecutive k (24 bytes, 8-byte aligned): the k-contiguous A operand of the adjoint product dPF = dVP P^T
            unsigned short* dstb = dVPb + (s0 + l15) * ldk + (v0 + 4 * g) * 3;
            unsigned short hb[12];
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int d = 0; d < 3; ++d) hb[r * 3 + d] = __builtin_bit_cast(unsigned short, (__bf16)dvp[d][r]);
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                uint2 pk;
                pk.x = (unsigned)hb[4 * q] | ((unsigned)hb[4 * q + 1] << 16);
                pk.y = (unsigned)hb[4 * q + 2] | ((unsigned)hb[4 * q + 3] << 16);
                reinterpret_cast<uint2*>(dstb)[q] = pk;
            }
        } else {
        // d vp (transposed store: row 3v+d, 16 consecutive samples per 64-byte segment)
        float* dst = dVPt + ((v0 + 4 * g) * 3) * ldn + s0 + l15;
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int d = 0; d < 3; ++d) dst[(r * 3 + d) * ldn] = dvp[d][r];
        }
        if constexpr (ADJ3) {                                    // (behind the dvp store: its 12 registers are free)
            __builtin_amdgcn_sched_barrier(0);
            load_adjoint_weights();
            adj3_pieces();
            __builtin_amdgcn_sched_barrier(0);
            adj3_row(0);
            __builtin_amdgcn_sched_barrier(0);
            adj3_prefetch();
            adj3_row(1);
            __builtin_amdgcn_sched_barrier(0);
            adj3_row(2);
            __builtin_amdgcn_sched_barrier(0);
        }
        /*prof:c3*/
    }
    if constexpr (SPH && ADJ3) {
        // (the adjoint accumulated scale(W) scale(dT) x the sums: W images x 2^14, dT pieces x 2^12 -- back, exactly)
        constexpr float inv = 1.f / (16384.f * 4096.f);
#pragma unroll
        for (int e = 0; e < 12; ++e)
#pragma unroll
            for (int t = 0; t < 2; ++t) accdA[e][t] *= inv;
    }

    // ---- cross-wave reduction of dA through LDS (the staged sample data is dead now)
    __syncthreads();
    float* scr = lds;                                   // 2 x [16 samples][289] floats (odd stride: a half-wave's
                                                        // 16 samples x 2 joint groups land in 32 different banks)
    auto put = [&](int slot) {
#pragma unroll
        for (int e = 0; e < 12; ++e)
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int j = 16 * t + 4 * g + r;
                    if (j < 24) scr[(slot * 16 + l15) * 289 + j * 12 + e] = accdA[e][t][r];
                }
    };
    auto take = [&](int slot) {
#pragma unroll
        for (int e = 0; e < 12; ++e)
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int j = 16 * t + 4 * g + r;
                    if (j < 24) accdA[e][t][r] += scr[(slot * 16 + l15) * 289 + j * 12 + e];
                }
    };
    if (wid >= 2) put(wid - 2);
    __syncthreads();
    if (wid < 2) take(wid);
    __syncthreads();
    if (wid == 1) put(0);
    __syncthreads();
    // The nr = gridDim.y blocks of a sample group hold partial sums over their vertex ranges.  Each publishes
    // its partial as a coalesced register image (write-through stores), takes a ticket, and the LAST
    // arriver adds the partials and writes dA: a fixed order (deterministic), and none of the ~2 M
    // same-address fp32 atomics this used to cost (44 us at N = 2400, 95 us at N = 300).  All four waves of
    // the last arriver share the summation (wave w takes ranges w, w+4, ...: with 26 ranges per group at a
    // one-instance shard a single wave spent ~35 us on 26 dependent memory round trips).
    // dA == NULL: DEFERRED combine -- every block leaves its partial image in the scratch and is done; nemo_v2v_combine
    // (a launch of its own, which the caller can run beside the blend-shape adjoint GEMM: nothing before the FK adjoint
    // needs dA) sums them.  Otherwise the group's last-arriving block does, below.
    const bool defer = dA == nullptr;
    if (wid == 0) {
        take(0);
        if (nr > 1 || defer) {
            float* part = parts + ((size_t)grp * maxc + slot) * (96 * 64);
#pragma unroll
            for (int e = 0; e < 12; ++e)
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        __hip_atomic_store(part + ((e * 2 + t) * 4 + r) * 64 + lane, accdA[e][t][r],
                                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // sc1 store
            if (!defer) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                if (lane == 0)
                    ticket_old = __hip_atomic_fetch_add(tickets + grp, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
    __syncthreads();
    const bool finish = !defer && (nr == 1 || ticket_old == nr - 1);      // block-uniform
    if (nr > 1 && finish) {
        // (the partials are read with device-scope loads -- served by the memory side, never by a stale line of this XCD's
        //  L2 -- instead of behind an acquire fence: `buffer_inv sc1` walks the whole L2)
        if (tid == 0) __hip_atomic_store(tickets + grp, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        // Wave w owns image rows [24 w, 24 w + 24) of EVERY partial (complete sums, no cross-wave pass afterwards) and
        // keeps eight partials -- 192 coalesced loads -- in flight: three memory round trips for the 24 partials of a
        // one-instance shard, where a wave summing whole partials one after the other took six (+27 us on the
        // launch's tail, tools/mesh_timeline.py).  Partials are added in range order: deterministic.
        const float* base = parts + (size_t)grp * maxc * (96 * 64) + (size_t)(24 * wid) * 64 + lane;
        float acc[24];
#pragma unroll
        for (int q = 0; q < 24; ++q) acc[q] = 0.f;
        for (int pz0 = 0; pz0 < nr; pz0 += 8) {
            float v[8][24];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const float* pp = base + (size_t)min(pz0 + u, nr - 1) * (96 * 64);
#pragma unroll
                for (int q = 0; q < 24; ++q) v[u][q] = __hip_atomic_load(pp + q * 64, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (pz0 + u < nr) {
#pragma unroll
                    for (int q = 0; q < 24; ++q) acc[q] += v[u][q];
                }
        }
        if (s0 + l15 < N) {
#pragma unroll
            for (int q = 0; q < 24; ++q) {
                const int qq = 24 * wid + q, e = qq >> 3, t = (qq >> 2) & 1, r = qq & 3;
                const int j = 16 * t + 4 * g + r;
                if (j < 24) dA[(s0 + l15) * 288 + j * 12 + e] = acc[q];
            }
        }
    } else if (finish && wid == 0 && s0 + l15 < N) {
#pragma unroll
        for (int e = 0; e < 12; ++e)
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int j = 16 * t + 4 * g + r;
                    if (j < 24) dA[(s0 + l15) * 288 + j * 12 + e] = accdA[e][t][r];
                }
    }
    /*prof:out*/
    }   // segment
    // L1 sum: every block publishes its partial (write-through), the LAST block to arrive adds all of them in block
    // order in float64 and accumulates the result into the loss slot -- one writer per launch, launches of a
    // chunked batch are stream-ordered: the reported loss is bit-reproducible run to run and does not lose digits
    // at N = 262 144 (16 k fp32 atomics onto a 1e9 running sum used to cost 1e-4 relative).
    const float tot = block_sum(lsum, red);
    if (threadIdx.x == 0) {
        __hip_atomic_store(loss_parts + blockIdx.x, tot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        grid_last = __hip_atomic_fetch_add(grid_ticket, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (int)gridDim.x - 1;
    }
    __syncthreads();
    if (grid_last && threadIdx.x < 64) {
        double acc = 0.0;
        for (unsigned i = threadIdx.x; i < gridDim.x; i += 64)       // sc1 loads: served by L2, never a stale L1 line
            acc += (double)__hip_atomic_load(loss_parts + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off, 64);
        if (threadIdx.x == 0) {
            *loss_sum += (float)acc;
            __hip_atomic_store(grid_ticket, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // ready for the next launch
        }
    }
}


// Deferred combine of nemo_v2v_fused(dA = NULL): block = one group of 16 samples; wave w owns image rows [24 w, 24 w + 24)
// of every partial, eight partials in flight, added in range order (deterministic, the same order as the in-kernel path).
__global__ __launch_bounds__(256) void mesh_combine_kernel(long N, int nr, int maxc, const float* __restrict__ parts,
                                                           float* __restrict__ dA) {
    const int grp = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, l15 = lane & 15, g = lane >> 4;
    const long s0 = (long)grp * 16;
    const float* base = parts + (size_t)grp * maxc * (96 * 64) + (size_t)(24 * wid) * 64 + lane;
    float acc[24];
#pragma unroll
    for (int q = 0; q < 24; ++q) acc[q] = 0.f;
    for (int pz0 = 0; pz0 < nr; pz0 += 8) {
        float v[8][24];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const float* pp = base + (size_t)min(pz0 + u, nr - 1) * (96 * 64);
#pragma unroll
            for (int q = 0; q < 24; ++q) v[u][q] = pp[q * 64];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (pz0 + u < nr) {
#pragma unroll
                for (int q = 0; q < 24; ++q) acc[q] += v[u][q];
            }
    }
    if (s0 + l15 < N) {
#pragma unroll
        for (int q = 0; q < 24; ++q) {
            const int qq = 24 * wid + q, e = qq >> 3, t = (qq >> 2) & 1, r = qq & 3;
            const int j = 16 * t + 4 * g + r;
            if (j < 24) dA[(s0 + l15) * 288 + j * 12 + e] = acc[q];
        }
    }
}

// Temporal smoothness of the output joints, HuMoR's joints3d_smooth_loss
// (humor/humor/fitting/fitting_loss.py:366-370): 0.5 * sum_{v,t,j} |J[v,t+1,j] - J[v,t,j]|^2 over complete
// sequences laid out (view, frame).  One thread per (v, t, j): its share of the sum and the full gradient
// of its own joint (both neighbours), no atomics except the block sums of the scalar.
__global__ __launch_bounds__(256) void smooth_kernel(long V, long T, int J, const float* __restrict__ j3d,
                                                     float weight, float* __restrict__ scalar_out,
                                                     float* __restrict__ dj, NemoRed rr) {
    __shared__ float red[16];
    __shared__ int rflag;
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    float part = 0.f;
    if (i < V * T * J) {
        const long t = (i / J) % T;
        const float* c = j3d + i * 3;
        float g[3] = {0.f, 0.f, 0.f};
        if (t + 1 < T) {
            const float* n = c + (long)J * 3;
#pragma unroll
            for (int k = 0; k < 3; ++k) { const float d = n[k] - c[k]; part += 0.5f * d * d; g[k] -= d; }
        }
        if (t > 0) {
            const float* q = c - (long)J * 3;
#pragma unroll
            for (int k = 0; k < 3; ++k) g[k] += c[k] - q[k];
        }
        if (dj) { dj[i * 3] = weight * g[0]; dj[i * 3 + 1] = weight * g[1]; dj[i * 3 + 2] = weight * g[2]; }
    }
    const float tot = block_sum(part, red);
    nemo_red_scalar(tot, scalar_out, rr, (int)blockIdx.x, (int)gridDim.x, red, &rflag);      // (deterministic: common.h)
}

}  // namespace

// ------------------------------------------------------------------------------------------ C ABI
extern "C" int32_t nemo_fk_fwd(const nemo_ctx* ctx, int64_t rows, const float* R, float* A, float* Jp,
                               float* PF, int64_t ldpf, void* stream) {
    if (!ctx || rows < 0 || !R || !A || !Jp || (PF && ldpf < 207)) return NEMO_EINVAL;
    if (rows == 0) return NEMO_OK;
    static NemoAttrOnce attr_once;
    if (attr_once.need()) {
        HIPCHK(hipFuncSetAttribute((const void*)fk_fwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                   FK_LDS_BYTES));
    }
    hipLaunchKernelGGL(fk_fwd_kernel, dim3(nemo_cdiv(rows, FK_TB)), dim3(256), FK_LDS_BYTES,
                       (hipStream_t)stream, (long)rows, R, ctx->d_Jrest, ctx->kc, A, Jp, PF, (long)ldpf);
    NEMO_LAUNCH_CHECK();
    return NEMO_OK;
}

extern "C" int32_t nemo_fk_bwd(const nemo_ctx* ctx, int64_t rows, const float* R, const float* A,
                               const float* dA, const float* dJp, const float* dPF, int64_t lddpf, float* dR,
                               void* stream) {
    if (!ctx || rows < 0 || !R || !A || !dA || !dR || (dPF && lddpf < 207)) return NEMO_EINVAL;
    if (rows == 0) return NEMO_OK;
    static NemoAttrOnce attr_once;
    if (attr_once.need()) {
        HIPCHK(hipFuncSetAttribute((const void*)fk_bwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                   FK_LDS_BYTES));
    }
    hipLaunchKernelGGL(fk_bwd_kernel, dim3(nemo_cdiv(rows, FK_TB)), dim3(256), FK_LDS_BYTES,
                       (hipStream_t)stream, (long)rows, R, A, ctx->d_Jrest, ctx->kc, dA, dJp, dPF, (long)lddpf, dR);
    NEMO_LAUNCH_CHECK();
    return NEMO_OK;
}

static int kp_args(const nemo_ctx* ctx, KpArgs& a, int64_t N, int64_t V, int64_t T, const float* A,
                   const float* Jp, const float* Mq, int64_t ldq, const float* TR, int64_t ldt,
                   int32_t add_trans, const int64_t* view_idx, const int64_t* frame_idx, const float* cams,
                   const float* targets, const float* gt_size, float focal, float cx, float cy,
                   int32_t loss_type, int32_t mean_mode) {
    if (!ctx || N < 0 || V <= 0 || !A || !Jp || !view_idx || !cams) return NEMO_EINVAL;
    if (ctx->nq > 0 && (!Mq || ldq < ctx->nq * 72)) return NEMO_EINVAL;
    if (add_trans && (!TR || ldt < 3)) return NEMO_EINVAL;
    if (targets && !frame_idx) return NEMO_EINVAL;
    if (loss_type < 0 || loss_type > 5 || mean_mode < 0 || mean_mode > 1) return NEMO_EINVAL;
    if ((loss_type == 4 || loss_type == 5) && targets && !gt_size) return NEMO_EINVAL;
    a.N = N; a.V = V; a.T = T; a.A = A; a.Jp = Jp; a.Mq = Mq; a.TR = TR; a.w0 = ctx->d_w0;
    a.ldq = ldq; a.ldt = ldt; a.view_idx = view_idx; a.frame_idx = frame_idx; a.cams = cams;
    a.targets = targets; a.gt_size = gt_size; a.focal = focal; a.cx = cx; a.cy = cy;
    a.add_trans = add_trans; a.loss_type = loss_type; a.mean_mode = mean_mode;
    a.n_valid = nullptr;
    return NEMO_OK;
}

extern "C" int32_t nemo_kp_fwd(const nemo_ctx* ctx, int64_t N, int64_t V, int64_t T, const float* A,
                               const float* Jp, const float* Mq, int64_t ldq, const float* TR, int64_t ldt,
                               int32_t add_trans, const int64_t* view_idx, const int64_t* frame_idx,
                               const float* cams, const float* targets, const float* gt_size, float focal,
                               float cx, float cy, int32_t loss_type, int32_t mean_mode, float* j3d,
                               float* p2d, float* loss_all, float* view_acc, const int64_t* n_valid, void* stream) {
    KpArgs a;
    const int rc = kp_args(ctx, a, N, V, T, A, Jp, Mq, ldq, TR, ldt, add_trans, view_idx, frame_idx, cams,
                           targets, gt_size, focal, cx, cy, loss_type, mean_mode);
    if (rc) return rc;
    a.n_valid = n_valid;
    if (N == 0) return NEMO_OK;
    if (ctx->n_out > 32) return NEMO_EINVAL;
    const NemoRed rr = view_acc ? nemo_red_take(kp_dep_floats(N, V), 2) : NemoRed{nullptr, nullptr};
    hipLaunchKernelGGL(kp_fwd_kernel<32>, dim3(nemo_cdiv(N * 32, 256)), dim3(256), 0, (hipStream_t)stream, a,
                       ctx->kc, j3d, p2d, loss_all, view_acc, rr);
    NEMO_LAUNCH_CHECK();
    kp_launch_view_finish(rr, view_idx, N, V, view_acc, nullptr, 0, 2, (hipStream_t)stream);
    NEMO_LAUNCH_CHECK();
    return NEMO_OK;
}

extern "C" int32_t nemo_kp_finalize(int64_t V, int64_t n_out, int32_t W, int32_t mean_mode,
                                    const float* view_acc, float* scalar_out, float* norm, void* stream) {
    if (V <= 0 || n_out <= 0 || (W != 1 && W != 2) || !view_acc || !scalar_out || !norm) return NEMO_EINVAL;
    hipLaunchKernelGGL(kp_finalize_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, (long)V, (int)n_out,
                       (int)W, (int)mean_mode, view_acc, scalar_out, norm);
    NEMO_LAUNCH_CHECK();
    return NEMO_OK;
}

static int32_t kp_bwd_impl(const nemo_ctx* ctx, int64_t N, int64_t V, int64_t T, const float* A,
                               const float* Jp, const float* Mq, int64_t ldq, const float* TR, int64_t ldt,
                               int32_t add_trans, const int64_t* view_idx, const int64_t* frame_idx,
                               const float* cams, const float* targets, const float* gt_size, float focal,
                               float cx, float cy, int32_t loss_type, int32_t mean_mode,
                               const float* view_acc, const float* norm, float upstream, float* dA,
                               float* dJp, float* dMq, float* dTR, int64_t lddt, float* d_cams,
                               const float* dj3d_extra, const int64_t* n_valid, void* stream) {
    KpArgs a;
    const int rc = kp_args(ctx, a, N, V, T, A, Jp, Mq, ldq, TR, ldt, add_trans, view_idx, frame_idx, cams,
                           targets, gt_size, focal, cx, cy, loss_type, mean_mode);
    if (rc) return rc;
    a.n_valid = n_valid;
    if (!targets || !view_acc) return NEMO_EINVAL;            // (norm may be NULL: derived from view_acc)
    if (dA && (!dJp || (ctx->nq > 0 && !dMq))) return NEMO_EINVAL;
    if (N == 0) return NEMO_OK;
    if (ctx->n_out > 32) return NEMO_EINVAL;
    const NemoRed rr = d_cams ? nemo_red_take(kp_dep_floats(N, V), 2) : NemoRed{nullptr, nullptr};
    hipLaunchKernelGGL(kp_bwd_kernel<32>, dim3(nemo_cdiv(N * 32, 256)), dim3(256), 0, (hipStream_t)stream, a,
                       ctx->kc, view_acc, norm, upstream, dA, dJp, dMq, dTR, (long)lddt, d_cams, (int)ctx->nq, dj3d_extra,
                       (const int64_t*)nullptr, (float*)nullptr, (float*)nullptr, (float*)nullptr, (float*)nullptr, rr);
    NEMO_LAUNCH_CHECK();
    kp_launch_view_finish(rr, view_idx, N, V, nullptr, d_cams, 2, 11, (hipStream_t)stream);
    NEMO_LAUNCH_CHECK();
    return NEMO_OK;
}

extern "C" int32_t nemo_kp_bwd(const nemo_ctx* ctx, int64_t N, int64_t V, int64_t T, const float* A,
                               const float* Jp, const float* Mq, int64_t ldq, const float* TR, int64_t ldt,
                               int32_t add_trans, const int64_t* view_idx, const int64_t* frame_idx,
                               const float* cams, const float* targets, const float* gt_size, float focal,
                               float cx, float cy, int32_t loss_type, int32_t mean_mode,
                               const float* view_acc, const float* norm, float upstream, float* dA,
                               float* dJp, float* dMq, float* dTR, int64_t lddt, float* d_cams,
                               void* stream) {
    return kp_bwd_impl(ctx, N, V, T, A, Jp, Mq, ldq, TR, ldt, add_trans, view_idx, frame_idx, cams, targets,
                       gt_size, focal, cx, cy, loss_type, mean_mode, view_acc, norm, upstream, dA, dJp, dMq, dTR,
                       lddt, d_cams, nullptr, nullptr, stream);
}

extern "C" int32_t nemo_kp_bwd_ex(const nemo_ctx* ctx, int64_t N, int64_t V, int64_t T, const float* A,
                                  const float* Jp, const float* Mq, int64_t ldq, const float* TR, int64_t ldt,
                                  int32_t add_trans, const int64_t* view_idx, const int64_t* frame_idx,
                                  const float* cams, const float* targets, const float* gt_size, float focal,
                                  float cx, float cy, int32_t loss_type, int32_t mean_mode,
                                  const float* view_acc, const float* norm, float upstream, float* dA,
                                  float* dJp, float* dMq, float* dTR, int64_t lddt, float* d_cams,
                                  const float* dj3d_extra, const int64_t* n_valid, void* stream) {
    return kp_bwd_impl(ctx, N, V, T, A, Jp, Mq, ldq, TR, ldt, add_trans, view_idx, frame_idx, cams, targets,
                       gt_size, focal, cx, cy, loss_type, mean_mode, view_acc, norm, upstream, dA, dJp, dMq, dTR,
                       lddt, d_cams, dj3d_extra, n_valid, stream);
}

extern "C" int32_t nemo_kp_fwd_bwd(const nemo_ctx* ctx, int64_t N, int64_t V, int64_t T, const float* A, const float* Jp,
                                   const float* Mq, int64_t ldq, const float* TR, int64_t ldt, int32_t add_trans,
                                   const int64_t* view_idx, const int64_t* frame_idx, const float* cams, const float* targets,
                                   const float* gt_size, float focal, float cx, float cy, int32_t loss_type, int32_t mean_mode,
                                   const int64_t* view_count, float upstream, float* j3d, float* p2d, float* loss_all,
                                   float* view_acc, float* dA, float* dJp, float* dMq, float* dTR, int64_t lddt,
                                   float* d_cams, const int64_t* n_valid, void* stream) {
    KpArgs a;
    const int rc = kp_args(ctx, a, N, V, T, A, Jp, Mq, ldq, TR, ldt, add_trans, view_idx, frame_idx, cams,
                           targets, gt_size, focal, cx, cy, loss_type, mean_mode);
    if (rc) return rc;
    a.n_valid = n_valid;
    if (!targets || !view_count || !view_acc) return NEMO_EINVAL;
    if (dA && (!dJp || (ctx->nq > 0 && !dMq))) return NEMO_EINVAL;
    if (N == 0) return NEMO_OK;
    if (ctx->n_out > 32) return NEMO_EINVAL;
    const NemoRed rr = nemo_red_take(kp_dep_floats(N, V), 2);
    hipLaunchKernelGGL((kp_bwd_kernel<32, true>), dim3(nemo_cdiv(N * 32, 256)), dim3(256), 0, (hipStream_t)stream, a,
                       ctx->kc, (const float*)nullptr, (const float*)nullptr, upstream, dA, dJp, dMq, dTR, (long)lddt, d_cams,
                       (int)ctx->nq, (const float*)nullptr, view_count, j3d, p2d, loss_all, view_acc, rr);
    NEMO_LAUNCH_CHECK();
    kp_launch_view_finish(rr, view_idx, N, V, view_acc, d_cams, 0, d_cams ? 11 : 2, (hipStream_t)stream);
    NEMO_LAUNCH_CHECK();
    return NEMO_OK;
}

extern "C" int32_t nemo_smooth_fwd_bwd(int64_t V, int64_t T, int64_t J, const float* j3d, float weight,
                                       float* scalar_out, float* dj3d, void* stream) {
    if (V <= 0 || T <= 0 || J <= 0 || !j3d || !scalar_out) return NEMO_EINVAL;
    hipLaunchKernelGGL(smooth_kernel, dim3(nemo_cdiv(V * T * J, 256)), dim3(256), 0, (hipStream_t)stream, (long)V,
                       (long)T, (int)J, j3d, weight, scalar_out, dj3d, nemo_red_take((size_t)nemo_cdiv(V * T * J, 256), 1));
    NEMO_LAUNCH_CHECK();
    return NEMO_OK;
}

extern "C" int32_t nemo_project(int64_t N, int64_t Jn, int64_t V, const float* pts, const int64_t* view_idx,
                                const float* cams, float focal, float cx, float cy, float* p2d,
                                void* stream) {
    if (N < 0 || Jn <= 0 || V <= 0 || !pts || !view_idx || !cams || !p2d) return NEMO_EINVAL;
    if (N == 0) return NEMO_OK;
    hipLaunchKernelGGL(project_kernel, dim3(nemo_cdiv(N * Jn, 256)), dim3(256), 0, (hipStream_t)stream,
                       (long)N, (int)Jn, pts, view_idx, cams, focal, cx, cy, p2d);
    NEMO_LAUNCH_CHECK();
    return NEMO_OK;
}

extern "C" int32_t nemo_skin_vertices(const nemo_ctx* ctx, int64_t rows, const float* VP, int64_t ldvp,
                                      const float* A, const float* trans, int64_t ldt, float* verts,
                                      void* stream) {
    if (!ctx || rows < 0 || !VP || !A || !verts || ldvp < ctx->NV * 3) return NEMO_EINVAL;
    if (rows == 0) return NEMO_OK;
    long gy = rows < 4096 ? rows : 4096;
    hipLaunchKernelGGL(skin_vertices_kernel, dim3(nemo_cdiv(ctx->NV, 256), (unsigned)gy), dim3(256), 0,
                       (hipStream_t)stream, (long)rows, ctx->NV, VP, (long)ldvp, A, ctx->d_Wt, trans,
                       (long)ldt, verts);
    NEMO_LAUNCH_CHECK();
    return NEMO_OK;
}

extern "C" int32_t nemo_v2v_skin_l1(const nemo_ctx* ctx, int64_t N, const float* VP, int64_t ldvp,
                                    const float* A, float* loss_sum, float* dVP, int64_t lddvp, float* dA,
                                    void* stream) {
    if (!ctx || N < 0 || !VP || !A || !loss_sum || !dVP || !dA || ldvp < ctx->NV * 3 || lddvp < ctx->NV * 3)
        return NEMO_EINVAL;
    if (N == 0) return NEMO_OK;
    hipLaunchKernelGGL(v2v_skin_l1_kernel<4>, dim3(nemo_cdiv(N, 4)), dim3(256), 0, (hipStream_t)stream, (long)N, ctx->NV, VP,
                       (long)ldvp, A, ctx->d_Wt, loss_sum, dVP, (long)lddvp, dA);
    NEMO_LAUNCH_CHECK();
    return NEMO_OK;
}

// Grid of the fused mesh kernel (see the kernel): RA range blocks per sample group over the chunks [0, CA)
// + nB blocks that each take the left-over range [CA, cpg) of up to k groups, G*RA + nB <= 512 co-resident
// blocks.  Cost in chunk units: the longest block, an extra segment charged MESH_SEG_OVH chunks (staging +
// partial-sum flush); ties go to fewer blocks.
struct MeshPlan { int G, cpg, RA, CA, nB; };
// Scratch layout of nemo_v2v_fused -- FIXED offsets, whatever the launch size: a caller may run launches of different
// sizes over one scratch buffer (the chunks of a large batch: 8192 + a ragged rest), and everything that must be zero
// between launches (the tickets) has to stay clear of what another launch size uses as plain data.
//   [0, 16)                         grid arrival ticket of the L1-sum reduction
//   [16, 16 + 4 MAX_BLOCKS)         one L1 partial per block
//   [.., + 4 MAX_GROUPS)            one arrival ticket per sample group
//   [MESH_HEADER_BYTES, ...)        partial dA images, groups x (RA + 1) x 96 x 64 floats
constexpr long MESH_MAX_BLOCKS = 65536, MESH_MAX_GROUPS = 65536;
constexpr long MESH_HEADER_BYTES = 16 + MESH_MAX_BLOCKS * 4 + MESH_MAX_GROUPS * 4;
constexpr double MESH_SEG_OVH = 1.0;
static MeshPlan mesh_plan(long groups, long ntiles) {
    const int G = (int)groups, C = (int)((ntiles + 3) / 4);
    if (const char* e = getenv("NEMO_MESH_PLAN")) {       // tuning / test aid: "RA,Lr,k" (not cached)
        int RA = 0, Lr = 0, k = 0;
        if (G >= 1 && sscanf(e, "%d,%d,%d", &RA, &Lr, &k) == 3 && RA >= 1 && Lr >= 0 && Lr < C && RA <= C - Lr &&
            (Lr == 0 || k >= 1))
            return MeshPlan{G, C, RA, C - Lr, Lr ? (G + k - 1) / k : 0};
    }
    static thread_local MeshPlan last{0, 0, 0, 0, 0};     // the search below is ~1e4 steps: keep the last answer
    if (last.G == G && last.cpg == C) return last;
    MeshPlan best{G, C, 1, C, 0};
    if (G > 512) return best;                             // one block per group, more than one resident wave
    if (G < 1) return best;
    // ties in the longest block go to the plan whose busiest CU (blocks b and b + 256 land on the same CU: round-robin
    // placement of one resident wave of blocks) carries least, then to fewer blocks
    auto busiest_cu = [&](int RA, int CA, int nB, int Lr) {
        const int q = CA / RA, rem = CA - q * RA, GA = G * RA;
        auto len = [&](int b) -> double {
            if (b < GA) return q + (b / G < rem ? 1 : 0);
            if (b >= GA + nB) return 0;
            const int j = b - GA;
            const int nseg = (int)(((long)G * (j + 1)) / nB) - (int)(((long)G * j) / nB);
            return nseg * (double)Lr + (nseg - 1) * MESH_SEG_OVH;
        };
        double m = 0;
        for (int c = 0; c < 256; ++c) { const double v = len(c) + len(c + 256); if (v > m) m = v; }
        return m;
    };
    double best_cost = 1e30, best_cu = 1e30;
    for (int RA = 1; RA * (long)G <= 512 && RA <= C; ++RA) {
        for (int k = 0; k <= 16; ++k) {                   // k = 0: no left-over blocks
            const int nB = k ? (G + k - 1) / k : 0;
            if ((long)G * RA + nB > 512) continue;
            const int keff = nB ? (G + nB - 1) / nB : 0;
            for (int Lr = k ? 1 : 0; Lr <= (k ? C - RA : 0); ++Lr) {
                const int CA = C - Lr;
                const double ca = (CA + RA - 1) / RA;                             // longest A range
                const double cb = k ? keff * Lr + (keff - 1) * MESH_SEG_OVH : 0;
                const double cost = (ca > cb ? ca : cb);
                if (cost > best_cost) continue;
                const double cu = busiest_cu(RA, CA, nB, Lr) + 1e-4 * ((long)G * RA + nB);
                if (cost < best_cost || cu < best_cu) { best_cost = cost; best_cu = cu; best = MeshPlan{G, C, RA, CA, nB}; }
            }
        }
    }
    last = best;
    return best;
}

extern "C" int64_t nemo_v2v_fused_ws_bytes(const nemo_ctx* ctx, int64_t N) {
    if (!ctx || N < 0) return -1;
    const long groups = (N + 15) / 16, ntiles = (ctx->NV + 15) / 16;
    if (groups == 0) return MESH_HEADER_BYTES;
    if (groups > MESH_MAX_GROUPS) return -1;
    const MeshPlan pl = mesh_plan(groups, ntiles);
    return MESH_HEADER_BYTES + groups * (long)(pl.RA + 1) * 96 * 64 * 4;
}

static int32_t v2v_fused_impl(const nemo_ctx* ctx, int kind, int64_t N, const float* PF2, int64_t ldpf,
                              const float* A2, float* loss_sum, float* dVPt, int64_t ldn, float* dA,
                              void* ws, int64_t ws_bytes, void* stream, unsigned short* dVPb = nullptr, int64_t ldk = 0, int64_t hplane = 0) {
    // kind: 0 = fp32, 1 = bf16 blend (+ split adjoint), 2 = fp32 with the blend's products on the bf16 pipe in three pieces
    const bool bf16 = kind == 1;
    if (!ctx || N < 0 || !PF2 || !A2 || !loss_sum || ldpf < 207) return NEMO_EINVAL;
    if (dVPb ? ((kind != 1 && kind != 2) || ldk < ctx->ldP || (ldk & 3) || (((uintptr_t)dVPb) & 7) ||
                (kind == 2 && hplane >= 0 && (hplane < ((N + 15) / 16) * 16 * ldk || (hplane & 3))) ||
                (kind == 2 && hplane < 0 && (ldk < 64 * ((3 * ctx->NVp + 31) / 32) || (ldk & 7))))
             : (!dVPt || ldn < ((N + 15) / 16) * 16))
        return NEMO_EINVAL;                                    // (dA == NULL: deferred combine, nemo_v2v_combine)
    if (N == 0) return NEMO_OK;
    // bf16: bf16 blend + split-precision vertex->joint adjoint AND split-precision skinning on the bf16 pipe (kernel MODE 2, the
    // default since round 5: re-measured after the LDS alignment fix of round 4 it is 4 - 7 % shorter than MODE 3 with sparse VALU
    // skinning -- 434 against 465 us per launch at 40 x 300, C3 step 2.34 -> 2.28 ms -- because the bf16 kernel is VALU-issue bound,
    // profiles/r05_pmc_mesh_b16.md).  NEMO_MESH_SPLIT=3: MODE 3 (fp32 skinning: sparse on the VALU or dense on the fp32 pipe), an
    // A/B aid.  MODE 1 (bf16 blend only) was measured in round 3 and is no longer instantiated.
    static const int split_env = getenv("NEMO_MESH_SPLIT") ? atoi(getenv("NEMO_MESH_SPLIT")) : 2;
    // kind 2 (fp32-equivalent split precision): two fp16 pieces per operand (MODE 5); NEMO_MESH_PIECES=3: three bf16 pieces (MODE 4, the
    // first form of the round: 360 against ... us per 8 x 300 launch), an A/B aid
    static const int pieces_env = getenv("NEMO_MESH_PIECES") ? atoi(getenv("NEMO_MESH_PIECES")) : 2;
    // Range guard: the fp16 pieces of the blended vertices hold |vp| 2^12 < 2^15.9 only while the body model's bound allows it
    // (nemo_ctx_split_ok, set by nemo_ctx_create / nemo_ctx_set_betas); beyond it the three-bf16-piece form runs -- and the caller
    // that asked for fp16 piece planes of d vp is refused (the engine asks nemo_ctx_split_ok first)
    if (kind == 2 && dVPb && !ctx->split_ok) return NEMO_EINVAL;
    int mode = bf16 ? (split_env == 3 ? 3 : 2) : kind == 2 ? (((pieces_env == 3 || !ctx->split_ok) && !dVPb) ? 4 : 5) : 0;   // (fp16 planes out: MODE 5 only)
    // MODE 6 (round 6, the default of kind 2; NEMO_MESH_SKIN=sparse: MODE 5): MODE 5 with both skinnings as fp16 split-precision MFMAs --
    // the dense 24-joint product of lbs.py:236-241 with the weights as three fp16 pieces (exact) and the transforms as two, four piece
    // products -- instead of the <= 4 non-zero weights on the VALU: 357 -> 344 us per 8 x 300 launch; inside the range guard of the
    // transforms' pieces only (nemo_ctx_skin_mfma_ok: 2^12 |A| < 2^15.9)
    static const bool skin_mfma = !(getenv("NEMO_MESH_SKIN") != nullptr && !strcmp(getenv("NEMO_MESH_SKIN"), "sparse"));
    if (mode == 5 && skin_mfma && ctx->skin_mfma_ok) mode = 6;
    const bool sparse = ctx->skin_sparse != 0 && mode != 2 && mode != 6;
    const int lds_bytes = mode == 6 ? (2 * 2 * 16 * MF_PFB / 2 + 2 * 2 * 16 * MF_AB / 2 + MF_TAIL) * (int)sizeof(float)
        : mode == 2 ? (2 * 16 * MF_PFB / 2 + 2 * 2 * 16 * MF_AB / 2 + MF_TAIL) * (int)sizeof(float)
        : ((mode == 4 ? 3 * 2 * 16 * MF_PFB / 2 : mode == 5 ? 2 * 2 * 16 * MF_PFB / 2 : bf16 ? 2 * 16 * MF_PFB / 2 : 2 * 16 * MF_PFS) +
           2 * 16 * (sparse ? MF_ASP : MF_AS) + MF_TAIL) *
              (int)sizeof(float);
    static NemoAttrOnce attr_once[7][2];
    if (attr_once[mode][sparse].need()) {
        const void* fn = mode == 6 ? (const void*)mesh_v2v_fused_kernel<6, false>
                       : mode == 5 ? (sparse ? (const void*)mesh_v2v_fused_kernel<5, true> : (const void*)mesh_v2v_fused_kernel<5, false>)
                       : mode == 4 ? (sparse ? (const void*)mesh_v2v_fused_kernel<4, true> : (const void*)mesh_v2v_fused_kernel<4, false>)
                       : mode == 2 ? (const void*)mesh_v2v_fused_kernel<2, false>
                       : mode == 3 ? (sparse ? (const void*)mesh_v2v_fused_kernel<3, true> : (const void*)mesh_v2v_fused_kernel<3, false>)
                                   : (sparse ? (const void*)mesh_v2v_fused_kernel<0, true> : (const void*)mesh_v2v_fused_kernel<0, false>);
        HIPCHK(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes));
    }
    const int vec_stage = (ldpf % 4 == 0) && (((uintptr_t)PF2 | (uintptr_t)A2) & 15) == 0 && ldpf >= 208;
    const long groups = (N + 15) / 16, ntiles = (ctx->NV + 15) / 16;
    const MeshPlan pl = mesh_plan(groups, ntiles);
    // scratch: one arrival ticket per sample group (zero at allocation, returned to zero by the kernel) and
    // the partial dA images of the blocks that overlap a group
    const long blocks = (long)pl.G * pl.RA + pl.nB;
    if (!ws || (((uintptr_t)ws) & 15) || blocks > MESH_MAX_BLOCKS || groups > MESH_MAX_GROUPS ||
        ws_bytes < MESH_HEADER_BYTES + groups * (long)(pl.RA + 1) * 96 * 64 * 4)
        return NEMO_EINVAL;
    char* wsb = reinterpret_cast<char*>(ws);
    int* grid_ticket = reinterpret_cast<int*>(wsb);
    float* loss_parts = reinterpret_cast<float*>(wsb + 16);
    int* tickets = reinterpret_cast<int*>(wsb + 16 + MESH_MAX_BLOCKS * 4);
    float* parts = reinterpret_cast<float*>(wsb + MESH_HEADER_BYTES);
    const unsigned short* WADJ = (mode == 5 || mode == 6) ? ctx->d_Wadjh : mode == 4 ? ctx->d_Wadj3 : ctx->d_Wadj;
    const unsigned short* WSK = mode == 6 ? ctx->d_Wskh : ctx->d_Wsk;
#define MESH_LAUNCH(M, SP, PP, LDP) hipLaunchKernelGGL((mesh_v2v_fused_kernel<M, SP>), dim3((unsigned)blocks), dim3(256), lds_bytes, \
        (hipStream_t)stream, (long)N, ctx->NV, PF2, (long)ldpf, A2, PP, LDP, ctx->d_v_shaped, ctx->d_W, pl.G, pl.cpg, pl.RA,  \
        pl.CA, pl.nB, vec_stage, loss_sum, dVPt, (long)ldn, dA, parts, tickets, loss_parts, grid_ticket, dVPb, (long)ldk,    \
        WSK, WADJ, ctx->d_Wsp_w, ctx->d_Wsp_j, ctx->sph_scale, (long)hplane)
    if (mode == 6) MESH_LAUNCH(6, false, reinterpret_cast<const float*>(ctx->d_posedirs_sph), ctx->NVp);
    else if (mode == 5 && sparse) MESH_LAUNCH(5, true, reinterpret_cast<const float*>(ctx->d_posedirs_sph), ctx->NVp);
    else if (mode == 5) MESH_LAUNCH(5, false, reinterpret_cast<const float*>(ctx->d_posedirs_sph), ctx->NVp);
    else if (mode == 4 && sparse) MESH_LAUNCH(4, true, reinterpret_cast<const float*>(ctx->d_posedirs_sp3), ctx->NVp);
    else if (mode == 4) MESH_LAUNCH(4, false, reinterpret_cast<const float*>(ctx->d_posedirs_sp3), ctx->NVp);
    else if (mode == 2) MESH_LAUNCH(2, false, reinterpret_cast<const float*>(ctx->d_posedirs_bf16), ctx->NVp);
    else if (mode == 3 && sparse) MESH_LAUNCH(3, true, reinterpret_cast<const float*>(ctx->d_posedirs_bf16), ctx->NVp);
    else if (mode == 3) MESH_LAUNCH(3, false, reinterpret_cast<const float*>(ctx->d_posedirs_bf16), ctx->NVp);
    else if (sparse) MESH_LAUNCH(0, true, ctx->d_posedirs, ctx->ldP);
    else MESH_LAUNCH(0, false, ctx->d_posedirs, ctx->ldP);
#undef MESH_LAUNCH
    NEMO_LAUNCH_CHECK();
    return NEMO_OK;
}

extern "C" int32_t nemo_v2v_combine(const nemo_ctx* ctx, int64_t N, float* dA, const void* ws, int64_t ws_bytes,
                                    void* stream) {
    if (!ctx || N < 0 || !dA || !ws) return NEMO_EINVAL;
    if (N == 0) return NEMO_OK;
    const long groups = (N + 15) / 16, ntiles = (ctx->NV + 15) / 16;
    if (groups > MESH_MAX_GROUPS) return NEMO_EINVAL;
    const MeshPlan pl = mesh_plan(groups, ntiles);             // the plan the fused launch of the same N used
    if (ws_bytes < MESH_HEADER_BYTES + groups * (long)(pl.RA + 1) * 96 * 64 * 4) return NEMO_EINVAL;
    const float* parts = reinterpret_cast<const float*>(reinterpret_cast<const char*>(ws) + MESH_HEADER_BYTES);
    hipLaunchKernelGGL(mesh_combine_kernel, dim3((unsigned)groups), dim3(256), 0, (hipStream_t)stream, (long)N,
                       pl.RA + (pl.CA < pl.cpg ? 1 : 0), pl.RA + 1, parts, dA);
    NEMO_LAUNCH_CHECK();
    return NEMO_OK;
}

extern "C" int32_t nemo_v2v_fused(const nemo_ctx* ctx, int64_t N, const float* PF2, int64_t ldpf,
                                  const float* A2, float* loss_sum, float* dVPt, int64_t ldn, float* dA,
                                  void* ws, int64_t ws_bytes, void* stream) {
    return v2v_fused_impl(ctx, 0, N, PF2, ldpf, A2, loss_sum, dVPt, ldn, dA, ws, ws_bytes, stream);
}

// fp32 in, fp32 out, fp32-equivalent arithmetic: the pose blend's products on the bf16 pipe with both operands in three bf16
// pieces (kernel MODE 4); same arguments and outputs as nemo_v2v_fused
extern "C" int32_t nemo_v2v_fused_split(const nemo_ctx* ctx, int64_t N, const float* PF2, int64_t ldpf,
                                         const float* A2, float* loss_sum, float* dVPt, int64_t ldn, float* dA,
                                         void* ws, int64_t ws_bytes, void* stream) {
    return v2v_fused_impl(ctx, 2, N, PF2, ldpf, A2, loss_sum, dVPt, ldn, dA, ws, ws_bytes, stream);
}

// nemo_v2v_fused_split whose d vp output is two fp16 piece planes of 2^12 d vp, NOT transposed: dVPh (16 * ceil(N / 16) rows x ldk >= the
// blend-shape row stride, plane 1 `plane` elements behind plane 0) -- the A operand of nemo_gemm_f16x2mem_adj
extern "C" int32_t nemo_v2v_fused_splitmem(const nemo_ctx* ctx, int64_t N, const float* PF2, int64_t ldpf, const float* A2,
                                           float* loss_sum, uint16_t* dVPh, int64_t ldk, int64_t plane, float* dA, void* ws,
                                           int64_t ws_bytes, void* stream) {
    if (!dVPh) return NEMO_EINVAL;
    return v2v_fused_impl(ctx, 2, N, PF2, ldpf, A2, loss_sum, nullptr, 0, dA, ws, ws_bytes, stream, dVPh, ldk, plane);
}

// nemo_v2v_fused_splitmem whose d vp output is ONE xp matrix (include/nemo_hip.h nemo_gemm_xp fmt 2: the two fp16 pieces of 2^12 d vp
// interleaved per k-block of 32): dVPx (16 * ceil(N / 16) rows x ldk >= nemo_xp_ld(2, 3 NVp)) -- the A operand of the blend-shape
// adjoint through nemo_gemm_xp (round 6)
extern "C" int32_t nemo_v2v_fused_splitxp(const nemo_ctx* ctx, int64_t N, const float* PF2, int64_t ldpf, const float* A2,
                                          float* loss_sum, uint16_t* dVPx, int64_t ldk, float* dA, void* ws, int64_t ws_bytes,
                                          void* stream) {
    if (!dVPx) return NEMO_EINVAL;
    return v2v_fused_impl(ctx, 2, N, PF2, ldpf, A2, loss_sum, nullptr, 0, dA, ws, ws_bytes, stream, dVPx, ldk, -1);
}

// bf16 variant whose d vp output is bf16 and NOT transposed: dVPb (16 * ceil(N / 16) rows x ldk >= 3 NVp, bf16) -- the
// k-contiguous operand nemo_gemm_bf16mem takes for the blend-shape adjoint (half the bytes written here and read there)
extern "C" int32_t nemo_v2v_fused_bf16mem(const nemo_ctx* ctx, int64_t N, const float* PF2, int64_t ldpf, const float* A2,
                                          float* loss_sum, uint16_t* dVPb, int64_t ldk, float* dA, void* ws, int64_t ws_bytes,
                                          void* stream) {
    return v2v_fused_impl(ctx, 1, N, PF2, ldpf, A2, loss_sum, nullptr, 0, dA, ws, ws_bytes, stream, dVPb, ldk);
}

extern "C" int32_t nemo_v2v_fused_bf16(const nemo_ctx* ctx, int64_t N, const float* PF2, int64_t ldpf,
                                       const float* A2, float* loss_sum, float* dVPt, int64_t ldn, float* dA,
                                       void* ws, int64_t ws_bytes, void* stream) {
    return v2v_fused_impl(ctx, 1, N, PF2, ldpf, A2, loss_sum, dVPt, ldn, dA, ws, ws_bytes, stream);
}
