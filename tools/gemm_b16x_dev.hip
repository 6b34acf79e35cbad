// Stand-alone numerics + timing harness of the large-tile bf16 GEMM (nemo_cvpr2023_amd/csrc/gemm_b16x.h).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize tools/gemm_b16x_dev.hip -o tools/gemm_b16x_dev
//   ./tools/gemm_b16x_dev check     every tile configuration x epilogue against a float64 host product (ragged M / N / K,
//                                   NaN-poisoned pads, K slices, += mode, tickets back at zero)
//   ./tools/gemm_b16x_dev time [M]  the bf16 chain's products at M rows (default 12001 = C3) per configuration, beside the
//                                   64 x 64 LDS-DMA kernel they replace; TFLOP/s; VERDICT r04 item 1's kill criterion
//   ./tools/gemm_b16x_dev trprobe   what ds_read_b64_tr_b16 returns for lane-linear addresses (documents the semantics the
//                                   next step -- products over NON-transposed activation copies -- would rely on)
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <vector>
#include <string>
#include <algorithm>
#include "../nemo_cvpr2023_amd/csrc/gemm_b16x.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

static unsigned short f2b(float f) {
    unsigned u; memcpy(&u, &f, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (unsigned short)((u >> 16) | 0x40);
    u += 0x7fffu + ((u >> 16) & 1);
    return (unsigned short)(u >> 16);
}
static float b2f(unsigned short b) { unsigned u = (unsigned)b << 16; float f; memcpy(&f, &u, 4); return f; }
static float frand() { return (float)(rand() % 2001 - 1000) / 1000.f; }
static long up8(long x) { return (x + 7) / 8 * 8; }

struct Feat { const char* name; bool c, cb, cbt, mask, colsum, bias; int act, out_mode, split; };

static int check() {
    struct P { long M, N, K; };
    const P probs[] = {{130, 70, 100}, {192, 256, 64}, {301, 147, 1000}, {513, 1000, 152}, {1000, 520, 1201}, {257, 300, 77}, {1, 9, 6}, {700, 1000, 1000}};
    const Feat feats[] = {
        {"C bias relu", true, false, false, false, false, true, 1, 0, 1},
        {"Cb CbT bias relu", false, true, true, false, false, true, 1, 0, 1},
        {"mask Cb CbT colsum", false, true, true, true, true, false, 0, 0, 1},
        {"mask C colsum", true, false, false, true, true, false, 0, 0, 1},
        {"C += split 3", true, false, false, false, false, false, 0, 1, 3},
        {"C Cb split 5 leaky", true, true, false, false, false, true, 2, 0, 5},
        {"C += split 11", true, false, false, false, false, false, 0, 1, 11},
    };
    float* ws; CK(hipMalloc(&ws, 256 << 20)); CK(hipMemset(ws, 0, 256 << 20));
    int bad = 0;
    for (const P& p : probs) {
        const long lda = up8(p.K) + 8, ldb = up8(p.K) + 16, ldc = p.N + 3, ldcb = up8(p.N) + 8, ldcbt = up8(p.M) + 8, ldm = up8(p.N) + 8;
        std::vector<unsigned short> hA(p.M * lda), hB(p.N * ldb), hMask(p.M * ldm);
        const unsigned short NANB = 0x7fc0;
        for (long m = 0; m < p.M; ++m) for (long k = 0; k < lda; ++k) hA[m * lda + k] = k < p.K ? f2b(frand()) : NANB;
        for (long n = 0; n < p.N; ++n) for (long k = 0; k < ldb; ++k) hB[n * ldb + k] = k < p.K ? f2b(frand()) : NANB;
        for (long m = 0; m < p.M; ++m) for (long n = 0; n < ldm; ++n) {
            const int q = rand() % 4;                       // > 0, == 0, < 0, -0
            hMask[m * ldm + n] = q == 0 ? f2b(0.f) : q == 1 ? f2b(-frand() * frand() - 0.1f) : q == 2 ? f2b(0.3f + fabsf(frand())) : (unsigned short)0x8000;
        }
        std::vector<float> hbias(p.N);
        for (auto& x : hbias) x = frand();
        std::vector<double> ref(p.M * p.N);
        for (long m = 0; m < p.M; ++m) for (long n = 0; n < p.N; ++n) {
            double s = 0;
            for (long k = 0; k < p.K; ++k) s += (double)b2f(hA[m * lda + k]) * b2f(hB[n * ldb + k]);
            ref[m * p.N + n] = s;
        }
        unsigned short *dA, *dB, *dMask, *dCb, *dCbT; float *dC, *dbias, *dcs;
        const long csrows = 2 * ((p.M + 63) / 64);
        CK(hipMalloc(&dA, hA.size() * 2)); CK(hipMalloc(&dB, hB.size() * 2)); CK(hipMalloc(&dMask, hMask.size() * 2));
        CK(hipMalloc(&dCb, p.M * ldcb * 2)); CK(hipMalloc(&dCbT, p.N * ldcbt * 2)); CK(hipMalloc(&dC, p.M * ldc * 4));
        CK(hipMalloc(&dbias, p.N * 4)); CK(hipMalloc(&dcs, csrows * p.N * 4));
        CK(hipMemcpy(dA, hA.data(), hA.size() * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(dB, hB.data(), hB.size() * 2, hipMemcpyHostToDevice));
        CK(hipMemcpy(dMask, hMask.data(), hMask.size() * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(dbias, hbias.data(), p.N * 4, hipMemcpyHostToDevice));
        for (int cfg = 0; cfg < 9; ++cfg)
            for (const Feat& f : feats) {
                if (cfg == 4 || cfg == 7) continue;
                if (f.split > 1 && p.K < 64 * f.split) continue;
                b16x::Args g{};
                g.A = dA; g.B = dB; g.M = p.M; g.N = p.N; g.K = p.K; g.lda = lda; g.ldb = ldb;
                g.C = f.c ? dC : nullptr; g.ldc = ldc; g.out_mode = f.out_mode; g.bias = f.bias ? dbias : nullptr; g.act = f.act; g.alpha = 0.5f;
                g.mask16 = f.mask ? dMask : nullptr; g.ldmask16 = ldm; g.mask_mode = f.mask ? 1 : 0;
                g.Cb = f.cb ? dCb : nullptr; g.ldcb = ldcb; g.CbT = f.cbt ? dCbT : nullptr; g.ldcbt = ldcbt;
                g.colsum = f.colsum ? dcs : nullptr; g.ldcs = p.N;
                g.counters = reinterpret_cast<int*>(ws); g.slabs = ws + 4096;
                if (!b16x::plan(g, cfg, f.split)) { printf("plan failed\n"); ++bad; continue; }
                std::vector<float> hC(p.M * ldc, f.out_mode ? 1.f : NAN);
                CK(hipMemcpy(dC, hC.data(), hC.size() * 4, hipMemcpyHostToDevice));
                CK(hipMemset(dCb, 0x11, p.M * ldcb * 2)); CK(hipMemset(dCbT, 0x11, p.N * ldcbt * 2)); CK(hipMemset(dcs, 0xff, csrows * p.N * 4));
                if (b16x::launch_cfg(cfg, g, 0) != hipSuccess) { printf("launch failed\n"); ++bad; continue; }
                CK(hipDeviceSynchronize());
                CK(hipGetLastError());
                std::vector<unsigned short> hCb(p.M * ldcb), hCbT(p.N * ldcbt);
                std::vector<float> hcs(csrows * p.N);
                CK(hipMemcpy(hC.data(), dC, hC.size() * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(hCb.data(), dCb, hCb.size() * 2, hipMemcpyDeviceToHost));
                CK(hipMemcpy(hCbT.data(), dCbT, hCbT.size() * 2, hipMemcpyDeviceToHost)); CK(hipMemcpy(hcs.data(), dcs, hcs.size() * 4, hipMemcpyDeviceToHost));
                double errc = 0, errb = 0, errt = 0, errs = 0, scale = 0;
                long nbadcb = 0;
                bool pads = true;
                std::vector<double> cs(csrows * p.N, 0.0);
                for (long m = 0; m < p.M; ++m) for (long n = 0; n < p.N; ++n) {
                    double v = 0.5 * ref[m * p.N + n] + (f.bias ? hbias[n] : 0.f);
                    if (f.act == 1) v = v > 0 ? v : 0; else if (f.act == 2) v = v > 0 ? v : 0.01 * v;
                    if (f.mask) { const float mv = b2f(hMask[m * ldm + n]); v = mv > 0.f ? v : 0; }
                    scale = fmax(scale, fabs(v));
                    cs[(m / 32) * p.N + n] += v;
                    if (f.c) { const double d = fabs((double)hC[m * ldc + n] - (f.out_mode ? 1.0 : 0.0) - v); if (!(d <= errc)) errc = d; }
                    if (f.cb) { const double d = fabs((double)b2f(hCb[m * ldcb + n]) - v) / (fabs(v) + 1e-2); if (!(d <= errb)) errb = d;
                                if (d > 0.1 && nbadcb++ < 12) printf("    Cb[%ld][%ld] = %g (bits %04x), want %g\n", m, n, b2f(hCb[m * ldcb + n]), hCb[m * ldcb + n], v); }
                    if (f.cbt) { const double d = fabs((double)b2f(hCbT[n * ldcbt + m]) - v) / (fabs(v) + 1e-2); if (!(d <= errt)) errt = d; }
                }
                if (f.colsum) for (long b = 0; b < csrows; ++b) for (long n = 0; n < p.N; ++n) { const double d = fabs(hcs[b * p.N + n] - cs[b * p.N + n]); if (!(d <= errs)) errs = d; }
                // pads: C columns beyond N untouched; Cb columns [N, up8(N)) zero, beyond untouched; CbT likewise along m
                if (f.c) for (long m = 0; m < p.M; ++m) for (long n = p.N; n < ldc; ++n) { const float v = hC[m * ldc + n]; if (f.out_mode ? v != 1.f : v == v) pads = false; }
                if (f.cb) for (long m = 0; m < p.M; ++m) for (long n = p.N; n < ldcb; ++n) { const unsigned short v = hCb[m * ldcb + n]; if (n < up8(p.N) ? v != 0 : v != 0x1111) pads = false; }
                if (f.cbt) for (long n = 0; n < p.N; ++n) for (long m = p.M; m < ldcbt; ++m) { const unsigned short v = hCbT[n * ldcbt + m]; if (m < up8(p.M) ? v != 0 : v != 0x1111) pads = false; }
                int tk = 0;
                for (int t = 0; t < g.tiles_m * g.tiles_n; ++t) { int x; CK(hipMemcpy(&x, reinterpret_cast<int*>(ws) + t, 4, hipMemcpyDeviceToHost)); tk |= x; }
                const bool ok = errc <= 2e-5 * scale + 1e-6 && errb <= 4.5e-3 && errt <= 4.5e-3 && errs <= 1e-4 * scale * 32 + 1e-5 && pads && tk == 0;
                if (!ok) ++bad;
                if (nbadcb) printf("    %ld wrong Cb elements\n", nbadcb);
                printf("cfg %d (%dx256%s) M=%4ld N=%4ld K=%4ld %-20s errC %.2g errCb %.2g errCbT %.2g errcs %.2g (scale %.3g) pads %s tickets %d %s\n", cfg,
                       b16x::tile_bm(cfg), cfg >= 6 ? "+4L k64" : cfg >= 3 ? "+4L" : "", p.M, p.N, p.K, f.name, errc, errb, errt, errs, scale, pads ? "ok" : "BAD", tk, ok ? "ok" : "FAIL");
            }
        CK(hipFree(dA)); CK(hipFree(dB)); CK(hipFree(dMask)); CK(hipFree(dCb)); CK(hipFree(dCbT)); CK(hipFree(dC)); CK(hipFree(dbias)); CK(hipFree(dcs));
    }
    printf(bad ? "CHECK FAILED (%d)\n" : "CHECK OK\n", bad);
    return bad;
}

// the 64 x 64 LDS-DMA kernel these products ran on until round 4 (gemm_glds.h, BF16 == 2), same operands
static hipError_t run_old(long M, long N, long K, const unsigned short* A, long lda, const unsigned short* B, long ldb, float* C, long ldc,
                          unsigned short* Cb, long ldcb, unsigned short* CbT, long ldcbt, const unsigned short* mask16, long ldm,
                          float* colsum, int out_mode, int split, float* ws) {
    glds::Args g{};
    g.A = reinterpret_cast<const float*>(A); g.B = reinterpret_cast<const float*>(B); g.C = C; g.M = M; g.N = N; g.K = K / 2; g.lda = lda / 2; g.ldb = ldb / 2;
    g.ldc = ldc; g.alpha = 1.f; g.out_mode = out_mode; g.Cb = Cb; g.ldcb = ldcb; g.CbT = CbT; g.ldcbt = ldcbt;
    g.mask16 = mask16; g.ldmask16 = ldm; g.mask_mode = mask16 ? 1 : 0; g.colsum = colsum; g.ldcs = N;
    g.counters = reinterpret_cast<int*>(ws); g.slabs = ws + 4096;
    if (!glds::extents(0, 1, M, N, K / 2, lda / 2, ldb / 2, &g.a_bytes, &g.b_bytes)) return hipErrorInvalidValue;
    long kc = (g.K + split - 1) / split; kc = (kc + 31) / 32 * 32;
    g.k_chunk = kc; g.split = (int)((g.K + kc - 1) / kc);
    g.tiles_m = (int)((M + 63) / 64); g.tiles_n = (int)((N + 63) / 64); g.n_tiles = g.tiles_m * g.tiles_n; g.t0 = 0;
    return glds::launch<64, 64, 32, 32, 32, true, true, 3, true, 2>(g, g.n_tiles * g.split, 0);
}

static void timeit(long Mr) {
    struct Shape { const char* name; long M, N, K; bool cb, cbt, mask, colsum, c; int out_mode; };
    const Shape shapes[] = {
        {"hidden fwd (Cb, CbT)", Mr, 1000, 1000, true, true, false, false, false, 0},
        {"hidden fwd, eval (Cb)", Mr, 1000, 1000, true, false, false, false, false, 0},
        {"hidden dX (mask, Cb, CbT, colsum)", Mr, 1000, 1000, true, true, true, true, false, 0},
        {"layer-2 dX (mask, C fp32, colsum)", Mr, 1000, 1000, false, false, true, true, true, 0},
        {"head dX (mask, Cb, CbT, colsum)", Mr, 1000, 152, true, true, true, true, false, 0},
        {"plain C fp32", Mr, 1000, 1000, false, false, false, false, true, 0},
        {"hidden dW (C +=)", 1000, 1000, Mr, false, false, false, false, true, 1},
        // the epilogues alone (one K tile)
        {"K = 64: Cb", Mr, 1000, 64, true, false, false, false, false, 0},
        {"K = 64: Cb, CbT", Mr, 1000, 64, true, true, false, false, false, 0},
        {"K = 64: mask, Cb, CbT, colsum", Mr, 1000, 64, true, true, true, true, false, 0},
        {"K = 64: mask, Cb", Mr, 1000, 64, true, false, true, false, false, 0},
        {"K = 64: C fp32", Mr, 1000, 64, false, false, false, false, true, 0},
        {"K = 64: colsum only", Mr, 1000, 64, false, false, false, true, false, 0},
    };
    float* ws; CK(hipMalloc(&ws, 256 << 20)); CK(hipMemset(ws, 0, 256 << 20));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto time_it = [&](auto go) {
        go(); go(); CK(hipDeviceSynchronize());
        float best = 1e30f;
        for (int rep = 0; rep < 3; ++rep) {
            CK(hipEventRecord(e0));
            for (int r = 0; r < 20; ++r) go();
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            best = fminf(best, ms * 50.f);
        }
        return best;
    };
    for (const Shape& p : shapes) {
        const long lda = up8(p.K), ldb = up8(p.K), ldc = p.N, ldcb = up8(p.N), ldcbt = up8(p.M) + 8, ldm = up8(p.N);
        std::vector<unsigned short> hA(p.M * lda), hB(p.N * ldb), hM(p.M * ldm);
        for (auto& x : hA) x = f2b(frand());
        for (auto& x : hB) x = f2b(frand());
        for (auto& x : hM) x = f2b(frand());
        unsigned short *dA, *dB, *dMask, *dCb, *dCbT; float *dC, *dcs;
        CK(hipMalloc(&dA, hA.size() * 2)); CK(hipMalloc(&dB, hB.size() * 2)); CK(hipMalloc(&dMask, hM.size() * 2));
        CK(hipMalloc(&dCb, p.M * ldcb * 2)); CK(hipMalloc(&dCbT, p.N * ldcbt * 2)); CK(hipMalloc(&dC, p.M * ldc * 4)); CK(hipMalloc(&dcs, (p.M / 16 + 16) * p.N * 4));
        CK(hipMemcpy(dA, hA.data(), hA.size() * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(dB, hB.data(), hB.size() * 2, hipMemcpyHostToDevice));
        CK(hipMemcpy(dMask, hM.data(), hM.size() * 2, hipMemcpyHostToDevice)); CK(hipMemset(dC, 0, p.M * ldc * 4));
        const double gf = 2e-9 * p.M * p.N * p.K;
        printf("%-36s M=%6ld N=%5ld K=%6ld  %.2f GFLOP\n", p.name, p.M, p.N, p.K, gf);
        for (int cfg = 0; cfg < 9; ++cfg) {
            if (cfg == 4 || cfg == 7) continue;
            std::string line = std::string("   ") + std::to_string(b16x::tile_bm(cfg)) + "x256" + (cfg >= 6 ? "+4L k64:" : cfg >= 3 ? "+4L:" : ":");
            for (int split : {1, 2, 4, 6, 8, 10, 12, 16}) {
                b16x::Args g{};
                g.A = dA; g.B = dB; g.M = p.M; g.N = p.N; g.K = p.K; g.lda = lda; g.ldb = ldb; g.C = p.c ? dC : nullptr; g.ldc = ldc; g.out_mode = p.out_mode;
                g.alpha = 1.f; g.mask16 = p.mask ? dMask : nullptr; g.ldmask16 = ldm; g.mask_mode = p.mask ? 1 : 0; g.Cb = p.cb ? dCb : nullptr; g.ldcb = ldcb;
                g.CbT = p.cbt ? dCbT : nullptr; g.ldcbt = ldcbt; g.colsum = p.colsum ? dcs : nullptr; g.ldcs = p.N;
                g.counters = reinterpret_cast<int*>(ws); g.slabs = ws + 4096;
                if (!b16x::plan(g, cfg, split)) continue;
                const long blocks = (long)g.tiles_m * g.tiles_n * g.split;
                if (split > 1 && (p.K / split < 256 || blocks > 640 || p.colsum)) continue;
                if (g.split > 1 && blocks * b16x::tile_bm(cfg) * 256 * 4 > (240L << 20)) continue;
                const float us = time_it([&] { CK(b16x::launch_cfg(cfg, g, 0)); });
                char b[96]; snprintf(b, sizeof b, " s%d[%ldb]=%.1fus(%.0fTF)", g.split, blocks, us, gf / us * 1e3);
                line += b;
            }
            printf("%s\n", line.c_str());
        }
        {
            std::string line = "   old 64x64:";
            for (int split : {1, 2, 3, 4}) {
                if (split > 1 && (p.K < 4096 || p.colsum)) continue;
                const float us = time_it([&] { CK(run_old(p.M, p.N, p.K, dA, lda, dB, ldb, p.c ? dC : nullptr, ldc, p.cb ? dCb : nullptr, ldcb, p.cbt ? dCbT : nullptr, ldcbt,
                                                          p.mask ? dMask : nullptr, ldm, p.colsum ? dcs : nullptr, p.out_mode, split, ws)); });
                char b[96]; snprintf(b, sizeof b, " s%d=%.1fus(%.0fTF)", split, us, gf / us * 1e3);
                line += b;
            }
            printf("%s\n", line.c_str());
        }
        CK(hipFree(dA)); CK(hipFree(dB)); CK(hipFree(dMask)); CK(hipFree(dCb)); CK(hipFree(dCbT)); CK(hipFree(dC)); CK(hipFree(dcs));
    }
}

// ---- ds_read_b64_tr_b16: every lane reads at base + 8 * lane from an LDS block holding element index i at bf16 slot i
__global__ void trprobe_kernel(unsigned* out) {
    __shared__ __attribute__((aligned(16))) unsigned short blk[1024];
    for (int i = threadIdx.x; i < 1024; i += 64) blk[i] = (unsigned short)i;
    __syncthreads();
    const unsigned addr = (unsigned)reinterpret_cast<unsigned long long>(blk) + 8u * threadIdx.x;
    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
    u32x2 r;
    asm volatile("ds_read_b64_tr_b16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=&v"(r) : "v"(addr) : "memory");
    out[2 * threadIdx.x] = r[0];
    out[2 * threadIdx.x + 1] = r[1];
}
static void trprobe() {
    unsigned* d; CK(hipMalloc(&d, 128 * 4));
    hipLaunchKernelGGL(trprobe_kernel, dim3(1), dim3(64), 0, 0, d);
    CK(hipDeviceSynchronize());
    unsigned h[128]; CK(hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost));
    printf("ds_read_b64_tr_b16, lane address = base + 8 * lane, LDS bf16 slot i holds i:\n");
    for (int l = 0; l < 64; ++l)
        printf("lane %2d: %4u %4u %4u %4u%s", l, h[2 * l] & 0xffff, h[2 * l] >> 16, h[2 * l + 1] & 0xffff, h[2 * l + 1] >> 16, (l & 3) == 3 ? "\n" : "   ");
}

// ---- does the scalar offset of a raw buffer access take part in the bounds check?  (the kernels point rows beyond an
// operand out of bounds through it)
__global__ void oobprobe_kernel(const unsigned* buf, unsigned bytes, unsigned* out) {
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned*>(buf), 0, (int)bytes, 0x00020000);
    out[threadIdx.x] = __builtin_amdgcn_raw_buffer_load_b32(rs, 4 * (int)threadIdx.x, 0, 0);                    // in bounds
    out[64 + threadIdx.x] = __builtin_amdgcn_raw_buffer_load_b32(rs, 4 * (int)threadIdx.x, (int)bytes, 0);      // soffset = extent
    out[128 + threadIdx.x] = __builtin_amdgcn_raw_buffer_load_b32(rs, 4 * (int)threadIdx.x + (int)bytes, 0, 0);  // voffset beyond
    out[192 + threadIdx.x] = __builtin_amdgcn_raw_buffer_load_b32(rs, 4 * (int)threadIdx.x, (int)bytes - 128, 0);// straddles the end
}
static void oobprobe() {
    unsigned *d, *o; CK(hipMalloc(&d, 4096)); CK(hipMalloc(&o, 256 * 4));
    std::vector<unsigned> h(1024); for (int i = 0; i < 1024; ++i) h[i] = 1000 + i;
    CK(hipMemcpy(d, h.data(), 4096, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(oobprobe_kernel, dim3(1), dim3(64), 0, 0, d, 1024u, o);
    CK(hipDeviceSynchronize());
    unsigned r[256]; CK(hipMemcpy(r, o, sizeof r, hipMemcpyDeviceToHost));
    printf("raw buffer, extent 1024 B of a 4096 B allocation holding 1000 + i:\n");
    printf("  in bounds             lanes 0,1,63: %u %u %u\n", r[0], r[1], r[63]);
    printf("  soffset = extent      lanes 0,1,63: %u %u %u   (0 = the scalar offset is bounds-checked)\n", r[64], r[65], r[127]);
    printf("  voffset beyond        lanes 0,1,63: %u %u %u\n", r[128], r[129], r[191]);
    printf("  soffset = extent-128  lanes 0,31,32,63: %u %u %u %u   (lanes >= 32 are beyond)\n", r[192], r[223], r[224], r[255]);
}

int main(int argc, char** argv) {
    const std::string mode = argc > 1 ? argv[1] : "check";
    if (mode == "check") return check();
    if (mode == "trprobe") { trprobe(); return 0; }
    if (mode == "oobprobe") { oobprobe(); return 0; }
    timeit(argc > 2 ? atol(argv[2]) : 12001);
    return 0;
}
