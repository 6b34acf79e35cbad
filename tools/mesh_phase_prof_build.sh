#!/bin/bash
# builds nemo_cvpr2023_amd/libnemo_hip_abl.so = current smpl.hip + cycle-counter probes
cd "$(dirname "$0")/../nemo_cvpr2023_amd/csrc" && python - <<'PY'
s=open('smpl.hip').read()
def rep(a,b):
    global s
    assert a in s, a[:50]
    s=s.replace(a,b,1)
rep("template <bool BF16>\n__global__ __launch_bounds__(256, 2) void mesh_v2v_fused_kernel(","__device__ unsigned long long mesh_prof[8 * 1024];\ntemplate <bool BF16>\n__global__ __launch_bounds__(256, 2) void mesh_v2v_fused_kernel(")
rep("    for (long t = t_beg + wid; t < t_end; t += 4) {\n        const long v0 = t * 16;","    unsigned long long pr_blend = 0, pr_rec = 0, pr_orig = 0, pr_tiles = 0, pr_sk = 0, pr_va = 0, pr_ad = 0, pr_st = 0, pr_t0 = __builtin_readcyclecounter();\n    for (long t = t_beg + wid; t < t_end; t += 4) {\n        unsigned long long c0 = __builtin_readcyclecounter();\n        const long v0 = t * 16;")
rep("        // ---- reconstruction body, one output row c (4 transform entries) at a time\n        float vrec[3][4];","        unsigned long long c1 = __builtin_readcyclecounter();\n        // ---- reconstruction body, one output row c (4 transform entries) at a time\n        float vrec[3][4];")
rep("        // ---- original body: row c of the transform -> vertex coordinate c -> sign -> its share of dvp and","        unsigned long long c2 = __builtin_readcyclecounter();\n        // ---- original body: row c of the transform -> vertex coordinate c -> sign -> its share of dvp and")
rep("            float gs[4];\n","            unsigned long long q1 = __builtin_readcyclecounter();\n            float gs[4];\n")
rep("            f32x4 T4[4];\n#pragma unroll\n            for (int d = 0; d < 4; ++d)\n#pragma unroll\n                for (int r = 0; r < 4; ++r) T4[d][r] = 0.f;\n#pragma unroll\n            for (int kk = 0; kk < 6; ++kk)\n#pragma unroll\n                for (int d = 0; d < 4; ++d)       // 4 independent accumulators back to back\n                    T4[d] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[kk], A0","            unsigned long long q0 = __builtin_readcyclecounter();\n            f32x4 T4[4];\n#pragma unroll\n            for (int d = 0; d < 4; ++d)\n#pragma unroll\n                for (int r = 0; r < 4; ++r) T4[d][r] = 0.f;\n#pragma unroll\n            for (int kk = 0; kk < 6; ++kk)\n#pragma unroll\n                for (int d = 0; d < 4; ++d)       // 4 independent accumulators back to back\n                    T4[d] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[kk], A0")
rep("            if (c == 2) {\n                // the wave's NEXT tile","            unsigned long long q2 = __builtin_readcyclecounter();\n            pr_sk += q1 - q0; pr_va += q2 - q1;\n            if (c == 2) {\n                // the wave's NEXT tile")
rep("            __builtin_amdgcn_sched_barrier(0);\n        }\n        // d vp (transposed store","            __builtin_amdgcn_sched_barrier(0);\n            pr_ad += __builtin_readcyclecounter() - q2;\n        }\n        unsigned long long q3 = __builtin_readcyclecounter();\n        // d vp (transposed store")
rep("            for (int d = 0; d < 3; ++d) dst[(r * 3 + d) * ldn] = dvp[d][r];\n    }\n","            for (int d = 0; d < 3; ++d) dst[(r * 3 + d) * ldn] = dvp[d][r];\n        unsigned long long c3 = __builtin_readcyclecounter();\n        pr_blend += c1 - c0; pr_rec += c2 - c1; pr_orig += c3 - c2; pr_tiles += 1; pr_st += c3 - q3;\n    }\n    unsigned long long pr_t1 = __builtin_readcyclecounter();\n")
rep("    }   // segment\n","    if (threadIdx.x == 0 && blockIdx.x < 1024 && seg == 0) {\n        unsigned long long* o = mesh_prof + blockIdx.x * 8;\n        o[0] = pr_blend; o[1] = pr_rec; o[2] = pr_orig; o[3] = pr_tiles; o[4] = pr_sk; o[5] = pr_va; o[6] = pr_ad; o[7] = pr_st;\n    }\n    }   // segment\n")
rep("    extern __shared__ float lds[];\n    float* pfL = lds;","    const unsigned long long pr_k0 = __builtin_readcyclecounter();\n    extern __shared__ float lds[];\n    float* pfL = lds;")
s+='''
extern "C" int32_t nemo_debug_mesh_prof(unsigned long long* out) {
    return (int32_t)hipMemcpyFromSymbol(out, HIP_SYMBOL(mesh_prof), sizeof(unsigned long long) * 8 * 1024);
}
'''
open('smpl_prof.hip','w').write(s)
PY
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function -Wno-pass-failed -c smpl_prof.hip -o /tmp/smpl_prof.o 2>&1 | grep -E "error" -A5 | head; /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC gemm.o pose.o /tmp/smpl_prof.o prior.o -o ../libnemo_hip_abl.so; rm -f smpl_prof.hip
