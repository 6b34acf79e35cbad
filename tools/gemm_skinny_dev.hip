// Stand-alone numerics + timing harness of the skinny GEMM (nemo_cvpr2023_amd/csrc/gemm_skinny.h) beside the LDS-DMA
// kernel with in-launch split-K (gemm_glds.h) on the shapes of the one-instance shard.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize tools/gemm_skinny_dev.hip -o tools/gemm_skinny_dev
//   ./tools/gemm_skinny_dev check      every configuration against a float64 host product (ragged M / N / K)
//   ./tools/gemm_skinny_dev time [N]   microseconds per launch inside a replayed HIP graph of 20 launches
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <vector>
#include <string>
#include "../nemo_cvpr2023_amd/csrc/gemm_skinny.h"

using glds::Args;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

enum { K_1x4 = 0, K_2x8, K_2x4, K_1x8, K_22x8, K_22x8d3, K_22x16, K_1x4u, K_2x8d8, K_1x4d8, NSK,
       // wide tiles for the blend-shape adjoint (N = 207 -> one column tile of 224): only in the `adj` mode
       K_17x4d3, K_17x4d2, K_17x4d4, K_27x2d2, K_14x4d3, K_14x8d3, NALL };
static const char* sk_name[] = {"32x32 w4 d4", "32x64 w8 d4", "32x64 w4 d4", "32x32 w8 d4", "64x64 w8 d4", "64x64 w8 d3", "64x64 w4 d4",
                                "32x32 w4 d4 unaligned", "32x64 w8 d8", "32x32 w4 d8", "-",
                                "32x224 w4 d3", "32x224 w4 d2", "32x224 w4 d4", "64x224 w2 d2", "32x128 w4 d3", "32x128 w8 d3"};

template <bool AKC, bool BKC>
void launch_sk(int cfg, const Args& g, hipStream_t s) {
    switch (cfg) {
        case K_1x4: CK((skinny::launch<AKC, BKC, true, true, 1, 1, 4, 4>(g, s))); break;
        case K_2x8: CK((skinny::launch<AKC, BKC, true, true, 1, 2, 8, 4>(g, s))); break;
        case K_2x4: CK((skinny::launch<AKC, BKC, true, true, 1, 2, 4, 4>(g, s))); break;
        case K_1x8: CK((skinny::launch<AKC, BKC, true, true, 1, 1, 8, 4>(g, s))); break;
        case K_22x8: CK((skinny::launch<AKC, BKC, true, true, 2, 2, 8, 4>(g, s))); break;
        case K_22x8d3: CK((skinny::launch<AKC, BKC, true, true, 2, 2, 8, 3>(g, s))); break;
        case K_22x16: CK((skinny::launch<AKC, BKC, true, true, 2, 2, 4, 4>(g, s))); break;
        case K_1x4u: CK((skinny::launch<AKC, BKC, false, false, 1, 1, 4, 4>(g, s))); break;
        case K_2x8d8: CK((skinny::launch<AKC, BKC, true, true, 1, 2, 8, 8>(g, s))); break;
        case K_1x4d8: CK((skinny::launch<AKC, BKC, true, true, 1, 1, 4, 8>(g, s))); break;
        case K_17x4d3: CK((skinny::launch<AKC, BKC, true, true, 1, 7, 4, 3>(g, s))); break;
        case K_17x4d2: CK((skinny::launch<AKC, BKC, true, true, 1, 7, 4, 2>(g, s))); break;
        case K_17x4d4: CK((skinny::launch<AKC, BKC, true, true, 1, 7, 4, 4>(g, s))); break;
        case K_27x2d2: CK((skinny::launch<AKC, BKC, true, true, 2, 7, 2, 2>(g, s))); break;
        case K_14x4d3: CK((skinny::launch<AKC, BKC, true, true, 1, 4, 4, 3>(g, s))); break;
        case K_14x8d3: CK((skinny::launch<AKC, BKC, true, true, 1, 4, 8, 3>(g, s))); break;
    }
}

static Args make_args(int ta, int tb, long M, long N, long K, const float* A, long lda, const float* B, long ldb, float* C,
                      long ldc, const float* bias, int act, const float* mask, int mask_mode, int out_mode) {
    Args g{};
    g.A = A; g.B = B; g.C = C; g.bias = bias; g.mask = mask; g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldb = ldb; g.ldc = ldc;
    g.ldmask = ldc; g.act = act; g.mask_mode = mask_mode; g.out_mode = out_mode; g.alpha = 1.f; g.xcd_order = 0;
    if (!glds::extents(ta, tb, M, N, K, lda, ldb, &g.a_bytes, &g.b_bytes)) { printf("extents\n"); exit(1); }
    return g;
}

static float* g_ws = nullptr;          // tickets (16 KiB, zero) + slabs
static int g_split = 1;                // K slices across blocks for run_sk

void run_sk(int cfg, int ta, int tb, const Args& g0, hipStream_t s) {
    Args g = g0;
    if (g_split > 1) {
        long kc = (g.K + g_split - 1) / g_split; kc = (kc + 7) / 8 * 8;
        g.k_chunk = kc; g.split = (int)((g.K + kc - 1) / kc);
        g.counters = reinterpret_cast<int*>(g_ws); g.slabs = g_ws + 4096;
    } else { g.split = 1; g.k_chunk = (g.K + 7) / 8 * 8; }
    const bool akc = !ta, bkc = tb;
    if (akc && bkc) launch_sk<true, true>(cfg, g, s);
    else if (akc && !bkc) launch_sk<true, false>(cfg, g, s);
    else if (!akc && bkc) launch_sk<false, true>(cfg, g, s);
    else launch_sk<false, false>(cfg, g, s);
}

// the shipped 64x64 LDS-DMA kernel with `split` K slices
void run_glds(int ta, int tb, Args g, int split, float* ws, hipStream_t s) {
    g.counters = reinterpret_cast<int*>(ws); g.slabs = ws + 4096;
    long kc = (g.K + split - 1) / split; kc = (kc + 31) / 32 * 32;
    g.k_chunk = kc; g.split = (int)((g.K + kc - 1) / kc);
    g.tiles_m = (int)((g.M + 63) / 64); g.tiles_n = (int)((g.N + 63) / 64); g.n_tiles = g.tiles_m * g.tiles_n; g.t0 = 0;
    const int blocks = g.n_tiles * g.split;
    const bool akc = !ta, bkc = tb;
    hipError_t e;
    if (akc && bkc) e = glds::launch<64, 64, 32, 32, 32, true, true, 3, true>(g, blocks, s);
    else if (akc && !bkc) e = glds::launch<64, 64, 32, 32, 32, true, false, 3, true>(g, blocks, s);
    else if (!akc && bkc) e = glds::launch<64, 64, 32, 32, 32, false, true, 3, true>(g, blocks, s);
    else e = glds::launch<64, 64, 32, 32, 32, false, false, 3, true>(g, blocks, s);
    CK(e);
}

// the 64x64 LDS-DMA kernel with NST stages of 32 k (one block per CU from 6 stages on): the whole K panel of a
// one-instance shard's parameter gradient (K = 301) requested up front
template <int NST>
void run_glds_deep(int ta, int tb, Args g, float* ws, hipStream_t s) {
    g.counters = reinterpret_cast<int*>(ws); g.slabs = ws + 4096;
    g.k_chunk = (g.K + 31) / 32 * 32; g.split = 1;
    g.tiles_m = (int)((g.M + 63) / 64); g.tiles_n = (int)((g.N + 63) / 64); g.n_tiles = g.tiles_m * g.tiles_n; g.t0 = 0;
    const bool akc = !ta, bkc = tb;
    hipError_t e;
    if (akc && bkc) e = glds::launch<64, 64, 32, 32, 32, true, true, NST, true>(g, g.n_tiles, s);
    else if (akc && !bkc) e = glds::launch<64, 64, 32, 32, 32, true, false, NST, true>(g, g.n_tiles, s);
    else if (!akc && bkc) e = glds::launch<64, 64, 32, 32, 32, false, true, NST, true>(g, g.n_tiles, s);
    else e = glds::launch<64, 64, 32, 32, 32, false, false, NST, true>(g, g.n_tiles, s);
    CK(e);
}

struct Prob { int ta, tb; long M, N, K; };
static float frand() { return (float)(rand() % 2001 - 1000) / 1000.f; }

int check() {
    const Prob probs_all[] = {{0, 1, 130, 70, 100}, {0, 1, 64, 64, 32}, {0, 1, 301, 147, 1000}, {0, 0, 301, 105, 1000},
                          {0, 0, 257, 200, 147}, {1, 0, 147, 1000, 301}, {1, 0, 200, 105, 333}, {1, 1, 300, 207, 2070},
                          {1, 1, 77, 207, 515}, {0, 1, 513, 512, 63}, {0, 1, 1, 9, 5}, {0, 1, 33, 300, 8}, {1, 0, 40, 40, 7}, {1, 0, 1000, 1000, 12000}};
    std::vector<Prob> probs(std::begin(probs_all), std::end(probs_all));
    if (getenv("SK_DEBUG")) probs.assign(1, probs_all[13]);
    int bad = 0;
    CK(hipMalloc(&g_ws, 64 << 20)); CK(hipMemset(g_ws, 0, 64 << 20));
    for (int split : {1, 3, 16})
    for (int odd = 0; odd < 3; ++odd)         // 1: rows that are not 16-byte aligned (+1), 2: rows with no pad at all
    for (const Prob& p : probs) {
        long lda = p.ta ? (p.M + 3) / 4 * 4 + 4 : (p.K + 3) / 4 * 4 + 4, ldb = p.tb ? (p.K + 3) / 4 * 4 + 8 : (p.N + 3) / 4 * 4;
        if (odd == 1) { lda += 1; ldb += 1; }
        if (odd == 2) { lda = p.ta ? p.M : p.K; ldb = p.tb ? p.K : p.N; }
        const long ar = p.ta ? p.K : p.M, br = p.tb ? p.N : p.K, ldc = p.N + 3;
        std::vector<float> hA(ar * lda), hB(br * ldb), hC(p.M * ldc), hC0(p.M * ldc), hM(p.M * ldc), hbias(p.N);
        for (auto& x : hA) x = frand();
        for (auto& x : hB) x = frand();
        for (auto& x : hbias) x = frand();
        for (auto& x : hC0) x = frand();
        for (auto& x : hM) x = frand();
        // poison the pads with NaN: nothing outside the logical operands may reach a result
        for (long r = 0; r < ar; ++r) for (long c = (p.ta ? p.M : p.K); c < lda; ++c) hA[r * lda + c] = NAN;
        for (long r = 0; r < br; ++r) for (long c = (p.tb ? p.K : p.N); c < ldb; ++c) hB[r * ldb + c] = NAN;
        std::vector<double> ref(p.M * p.N);
        for (long m = 0; m < p.M; ++m) for (long n = 0; n < p.N; ++n) {
            double s = 0;
            for (long k = 0; k < p.K; ++k) s += (double)(p.ta ? hA[k * lda + m] : hA[m * lda + k]) * (p.tb ? hB[n * ldb + k] : hB[k * ldb + n]);
            ref[m * p.N + n] = s;
        }
        float *dA, *dB, *dC, *dM, *dbias;
        // exact-size allocations: an out-of-bounds read would fault
        CK(hipMalloc(&dA, hA.size() * 4)); CK(hipMalloc(&dB, hB.size() * 4)); CK(hipMalloc(&dC, hC.size() * 4));
        CK(hipMalloc(&dM, hC.size() * 4)); CK(hipMalloc(&dbias, p.N * 4));
        CK(hipMemcpy(dA, hA.data(), hA.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dB, hB.data(), hB.size() * 4, hipMemcpyHostToDevice));
        CK(hipMemcpy(dM, hM.data(), hM.size() * 4, hipMemcpyHostToDevice));
        CK(hipMemcpy(dbias, hbias.data(), p.N * 4, hipMemcpyHostToDevice));
        g_split = split;
        if (split > 1 && (p.K < 64 * split || odd == 1)) continue;
        if (split > 1 && ((p.M + 31) / 32) * ((p.N + 31) / 32) * split * 4096L > (48L << 20)) continue;   // slabs must fit g_ws
        if (getenv("SK_DEBUG"))
            printf("A %p..%p B %p..%p C %p..%p M %p ws %p..%p\n", dA, dA + hA.size(), dB, dB + hB.size(), dC, dC + hC.size(), dM, g_ws, g_ws + (16 << 20));
        for (int cfg = odd ? (int)K_1x4u : 0; cfg < NSK; ++cfg)
            for (int variant = 0; variant < 3; ++variant) {
                if (split > 1 && cfg != K_1x4 && cfg != K_2x8 && cfg != K_1x4u && cfg != K_22x8d3) continue;
                if (getenv("SK_DEBUG")) { printf("cfg %d variant %d split %d\n", cfg, variant, split); fflush(stdout); }
                // 0: C = relu(AB + bias)   1: C += AB masked by M > 0   2: C = AB (no bias)
                CK(hipMemcpy(dC, hC0.data(), hC.size() * 4, hipMemcpyHostToDevice));
                const Args g = make_args(p.ta, p.tb, p.M, p.N, p.K, dA, lda, dB, ldb, dC, ldc, variant == 0 ? dbias : nullptr,
                                         variant == 0 ? 1 : 0, variant == 1 ? dM : nullptr, variant == 1 ? 1 : 0, variant == 1 ? 1 : 0);
                run_sk(cfg, p.ta, p.tb, g, 0);
                CK(hipGetLastError());
                CK(hipDeviceSynchronize());
                CK(hipMemcpy(hC.data(), dC, hC.size() * 4, hipMemcpyDeviceToHost));
                double err = 0, scale = 0;
                bool pad_ok = true;
                for (long m = 0; m < p.M; ++m) {
                    for (long n = 0; n < p.N; ++n) {
                        double want = ref[m * p.N + n];
                        if (variant == 0) { want += hbias[n]; want = want > 0 ? want : 0; }
                        if (variant == 1) want = hC0[m * ldc + n] + (hM[m * ldc + n] > 0 ? want : 0);
                        const double d = fabs((double)hC[m * ldc + n] - want);
                        if (!(d <= err)) err = d;
                        scale = fmax(scale, fabs(ref[m * p.N + n]));
                    }
                    for (long n = p.N; n < ldc; ++n) pad_ok &= hC[m * ldc + n] == hC0[m * ldc + n];
                }
                const bool ok = err <= 2e-5 * scale + 1e-6 && pad_ok;
                if (!ok) ++bad;
                printf("%-22s s%-2d ta=%d tb=%d M=%4ld N=%4ld K=%5ld variant %d  max err %.3g (scale %.3g) %s%s\n", sk_name[cfg], split, p.ta, p.tb,
                       p.M, p.N, p.K, variant, err, scale, ok ? "ok" : "FAIL", pad_ok ? "" : " (wrote outside C)");
            }
        CK(hipFree(dA)); CK(hipFree(dB)); CK(hipFree(dC)); CK(hipFree(dM)); CK(hipFree(dbias));
    }
    {   // every ticket is back at zero
        std::vector<int> t(4096);
        CK(hipMemcpy(t.data(), g_ws, 4096 * 4, hipMemcpyDeviceToHost));
        for (int x : t) if (x) { ++bad; printf("ticket left at %d\n", x); break; }
    }
    g_split = 1;
    printf(bad ? "CHECK FAILED (%d)\n" : "CHECK OK\n", bad);
    return bad;
}

// streams `n` floats through every L2 (read + write back): what the operands of a GEMM inside the step look like --
// written by the previous kernel on other XCDs, not resident in the reader's L2
__global__ void evict_kernel(float* p, long n) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) p[i] += 1.f;
}
static float* g_evict = nullptr;
static bool g_cold = false;
static void evict(hipStream_t s) { hipLaunchKernelGGL(evict_kernel, dim3(2048), dim3(256), 0, s, g_evict, (long)(48 << 20)); }

// microseconds per launch: 20 launches captured into one graph, the graph replayed
template <class F>
static double graph_time(F&& go, hipStream_t s) {
    hipGraph_t graph; hipGraphExec_t exec;
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
    const int reps = 20;
    for (int r = 0; r < reps; ++r) { if (g_cold) evict(s); go(); }
    CK(hipStreamEndCapture(s, &graph));
    CK(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipGraphLaunch(exec, s)); CK(hipStreamSynchronize(s));
    double best = 1e30;
    for (int it = 0; it < 3; ++it) {
        CK(hipEventRecord(e0, s)); CK(hipGraphLaunch(exec, s)); CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        best = fmin(best, ms * 1e3 / reps);
    }
    CK(hipGraphExecDestroy(exec)); CK(hipGraphDestroy(graph)); CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1));
    return best;
}

__global__ void touch_kernel(float* p) { if (threadIdx.x == 0) p[blockIdx.x] = 1.f; }

void timeit(long Nb) {
    struct Shape { const char* name; int ta, tb; long M, N, K; };
    const Shape shapes[] = {{"mlp_hidden_fwd", 0, 1, Nb, 1000, 1000}, {"mlp_hidden_dx", 0, 0, Nb, 1000, 1000},
                            {"mlp_hidden_dw", 1, 0, 1000, 1000, Nb}, {"mlp_in_fwd", 0, 1, Nb, 1000, 105}, {"mlp_in_dx", 0, 0, Nb, 105, 1000}, {"mlp_in_dw", 1, 0, 1000, 105, Nb},
                            {"head_fwd", 0, 1, Nb, 147, 1000}, {"head_dx", 0, 0, Nb, 1000, 147}, {"head_dw", 1, 0, 147, 1000, Nb},
                            {"vposer_512", 0, 1, Nb, 512, 512}, {"vposer_512_dx", 0, 0, Nb, 512, 512}, {"vposer_in", 0, 1, Nb, 512, 63},
                            {"vposer_mulv", 0, 1, Nb, 64, 512}, {"vposer_dec_out", 0, 1, Nb, 126, 512},
                            {"mq", 0, 0, Nb, 792, 207}, {"dpf_kp", 0, 1, Nb, 207, 792}, {"blend_adjoint", 1, 1, Nb, 207, 20670}};
    hipStream_t s; CK(hipStreamCreate(&s));
    float* ws; CK(hipMalloc(&ws, 256 << 20)); CK(hipMemset(ws, 0, 256 << 20));
    g_cold = getenv("SK_COLD") != nullptr;
    if (g_cold) {
        CK(hipMalloc(&g_evict, 192 << 20)); CK(hipMemset(g_evict, 0, 192 << 20));
        g_cold = false;
        const double base = graph_time([&] { evict(s); }, s);
        g_cold = true;
        printf("COLD mode: a 192 MB read-modify-write sweep before every launch (%.1f us, included in every number below)\n", base);
    }
    printf("empty 256-block kernel in the same graph: %.2f us per launch\n", graph_time([&] { hipLaunchKernelGGL(touch_kernel, dim3(256), dim3(256), 0, s, ws + 8192); }, s));
    for (const Shape& p : shapes) {
        const long lda = p.ta ? (p.M + 15) / 16 * 16 : (p.K + 3) / 4 * 4, ldb = p.tb ? (p.K + 3) / 4 * 4 : (p.N + 3) / 4 * 4;
        const long ar = p.ta ? p.K : p.M, br = p.tb ? p.N : p.K, ldc = (p.N + 3) / 4 * 4;
        std::vector<float> hA(ar * lda), hB(br * ldb);
        for (auto& x : hA) x = frand();
        for (auto& x : hB) x = frand();
        float *dA, *dB, *dC;
        CK(hipMalloc(&dA, hA.size() * 4)); CK(hipMalloc(&dB, hB.size() * 4)); CK(hipMalloc(&dC, p.M * ldc * 4));
        CK(hipMemcpy(dA, hA.data(), hA.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dB, hB.data(), hB.size() * 4, hipMemcpyHostToDevice));
        const Args g = make_args(p.ta, p.tb, p.M, p.N, p.K, dA, lda, dB, ldb, dC, ldc, nullptr, 1, nullptr, 0, 0);
        printf("%-14s ta=%d tb=%d M=%5ld N=%5ld K=%5ld  (%.2f GFLOP, %.1f us at 157.3 TF)\n", p.name, p.ta, p.tb, p.M, p.N, p.K,
               2e-9 * p.M * p.N * p.K, 2e-6 * p.M * p.N * p.K / 157.3);
        std::string line = "   glds 64x64:";
        const long tiles = ((p.M + 63) / 64) * ((p.N + 63) / 64);
        for (int split : {1, 2, 3, 4, 6, 8, 13, 26}) {
            if (split > 1 && (p.K / 32 / split < 2 || tiles * split > 1024)) continue;
            char buf[64]; snprintf(buf, sizeof buf, " s%d=%.1f", split, graph_time([&] { run_glds(p.ta, p.tb, g, split, ws, s); }, s));
            line += buf;
        }
        printf("%s\n", line.c_str());
        if (p.K <= 700 && tiles <= 512) {
            printf("   glds 64x64 deep pipeline: NST4=%.1f NST6=%.1f NST9=%.1f\n", graph_time([&] { run_glds_deep<4>(p.ta, p.tb, g, ws, s); }, s),
                   graph_time([&] { run_glds_deep<6>(p.ta, p.tb, g, ws, s); }, s), graph_time([&] { run_glds_deep<9>(p.ta, p.tb, g, ws, s); }, s));
        }
        line = "   skinny:";
        for (int cfg = 0; cfg < NSK; ++cfg) {
            char buf[64]; snprintf(buf, sizeof buf, "  [%s] %.1f", sk_name[cfg], graph_time([&] { run_sk(cfg, p.ta, p.tb, g, s); }, s));
            line += buf;
        }
        printf("%s\n", line.c_str());
        if (p.K > 4096) {
            g_ws = ws;
            for (int cfg : {(int)K_1x8, (int)K_2x8, (int)K_22x8d3}) {
                line = std::string("   skinny [") + sk_name[cfg] + "] x K slices:";
                for (int split : {4, 8, 12, 13, 16, 24}) {
                    g_split = split;
                    char buf[64]; snprintf(buf, sizeof buf, " s%d=%.1f", split, graph_time([&] { run_sk(cfg, p.ta, p.tb, g, s); }, s));
                    line += buf;
                }
                printf("%s\n", line.c_str());
            }
            g_split = 1;
        }
        CK(hipFree(dA)); CK(hipFree(dB)); CK(hipFree(dC));
    }
}

// the blend-shape adjoint dPF = dVP P^T (TT, N = 207, K = 20 670) at M = Nb: wide-tile configurations x K slices, checked
// against the 64x64 LDS-DMA kernel's result
void adjoint(long Nb, int tb) {
    const int ta = 1;
    const long M = Nb, N = 207, K = 20670;
    hipStream_t s; CK(hipStreamCreate(&s));
    float* ws; CK(hipMalloc(&ws, 256 << 20)); CK(hipMemset(ws, 0, 256 << 20));
    g_ws = ws;
    const long lda = (M + 15) / 16 * 16, ldb = tb ? 20672 : 208, ldc = 208;
    std::vector<float> hA(K * lda), hB((tb ? N : K) * ldb);
    for (auto& x : hA) x = frand();
    for (auto& x : hB) x = frand();
    float *dA, *dB, *dC, *dR;
    CK(hipMalloc(&dA, hA.size() * 4)); CK(hipMalloc(&dB, hB.size() * 4)); CK(hipMalloc(&dC, M * ldc * 4)); CK(hipMalloc(&dR, M * ldc * 4));
    CK(hipMemcpy(dA, hA.data(), hA.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dB, hB.data(), hB.size() * 4, hipMemcpyHostToDevice));
    Args g = make_args(ta, tb, M, N, K, dA, lda, dB, ldb, dR, ldc, nullptr, 0, nullptr, 0, 0);
    run_glds(ta, tb, g, 4, ws, s);
    CK(hipStreamSynchronize(s));
    std::vector<float> ref(M * ldc), got(M * ldc);
    CK(hipMemcpy(ref.data(), dR, ref.size() * 4, hipMemcpyDeviceToHost));
    g.C = dC;
    printf("blend adjoint (B %s) M=%ld N=%ld K=%ld: %.2f GFLOP, %.1f us at 157.3 TF\n", tb ? "[n][k]" : "[k][n]", M, N, K, 2e-9 * M * N * K, 2e-6 * M * N * K / 157.3);
    std::string line = "   glds 64x64:";
    for (int split : {1, 2, 3, 4, 6, 8, 13}) {
        if (((M + 63) / 64) * 4 * split > 1024) continue;
        char buf[64]; snprintf(buf, sizeof buf, " s%d=%.1f", split, graph_time([&] { run_glds(ta, tb, g, split, ws, s); }, s));
        line += buf;
    }
    printf("%s\n", line.c_str());
    for (int cfg : {(int)K_1x8, (int)K_17x4d3, (int)K_17x4d2, (int)K_17x4d4, (int)K_27x2d2, (int)K_14x4d3, (int)K_14x8d3}) {
        const int tn = (int)((N + (cfg == K_1x8 ? 31 : cfg >= K_14x4d3 ? 127 : 223)) / (cfg == K_1x8 ? 32 : cfg >= K_14x4d3 ? 128 : 224));
        const int tm = (int)((M + (cfg == K_27x2d2 ? 63 : 31)) / (cfg == K_27x2d2 ? 64 : 32));
        line = std::string("   skinny [") + sk_name[cfg] + "] tiles " + std::to_string(tm * tn) + " x K slices:";
        for (int split : {1, 2, 3, 4, 6, 8, 10, 13, 16, 20, 25, 32}) {
            if ((long)tm * tn * split > 2048 || (long)tm * tn * split < 64) continue;
            g_split = split;
            CK(hipMemsetAsync(dC, 0, M * ldc * 4, s));
            run_sk(cfg, ta, tb, g, s);
            CK(hipStreamSynchronize(s));
            CK(hipMemcpy(got.data(), dC, got.size() * 4, hipMemcpyDeviceToHost));
            double err = 0, scale = 0;
            for (long m = 0; m < M; ++m) for (long n = 0; n < N; ++n) {
                err = fmax(err, fabs((double)got[m * ldc + n] - ref[m * ldc + n])); scale = fmax(scale, fabs((double)ref[m * ldc + n]));
            }
            char buf[96]; snprintf(buf, sizeof buf, " s%d=%.1f%s", split, graph_time([&] { run_sk(cfg, ta, tb, g, s); }, s), err <= 1e-4 * scale ? "" : "(WRONG)");
            line += buf;
        }
        printf("%s\n", line.c_str());
    }
    g_split = 1;
}

int main(int argc, char** argv) {
    const std::string mode = argc > 1 ? argv[1] : "check";
    if (mode == "check") return check();
    if (mode == "adj") { adjoint(argc > 2 ? atol(argv[2]) : 300, argc > 3 ? atoi(argv[3]) : 1); return 0; }
    timeit(argc > 2 ? atol(argv[2]) : 300);
    return 0;
}
