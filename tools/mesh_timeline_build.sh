#!/bin/bash
# builds nemo_cvpr2023_amd/libnemo_hip_tl.so = current smpl.hip + wall-clock stamps (s_memrealtime, 100 MHz) at the
# phase boundaries of the mesh kernel: entry | staged | tile loop done | group combine done | end.  tools/mesh_timeline.py
cd "$(dirname "$0")/../nemo_cvpr2023_amd/csrc" && python - <<'PY'
s=open('smpl.hip').read()
def rep(a,b,count=1):
    global s
    assert a in s, a[:60]
    s=s.replace(a,b,count)
rep("template <bool BF16>\n__global__ __launch_bounds__(256, 2) void mesh_v2v_fused_kernel(","__device__ unsigned long long mesh_tl[8 * 1024];\ntemplate <bool BF16>\n__global__ __launch_bounds__(256, 2) void mesh_v2v_fused_kernel(")
rep("    extern __shared__ float lds[];\n    float* pfL = lds;","    unsigned long long tl[8] = {__builtin_amdgcn_s_memrealtime(), 0, 0, 0, 0, 0, 0, 0};\n    extern __shared__ float lds[];\n    float* pfL = lds;")
# after staging: the __syncthreads() that precedes the accumulator zeroing
i=s.index("            for (int r = 0; r < 4; ++r) accdA[e][t][r] = 0.f;\n    const float* pf0 = pfL")
j=s.rfind("    __syncthreads();\n", 0, i)
s=s[:j]+"    __syncthreads();\n    if (seg == 0) tl[1] = __builtin_amdgcn_s_memrealtime();\n"+s[j+len("    __syncthreads();\n"):]
rep("    if (wid >= 2) put(wid - 2);\n    __syncthreads();\n    if (wid < 2) take(wid);","    if (seg == 0) tl[2] = __builtin_amdgcn_s_memrealtime();\n    if (wid >= 2) put(wid - 2);\n    __syncthreads();\n    if (wid < 2) take(wid);")
rep("    __shared__ int ticket_old;\n","    if (seg == 0) tl[5] = __builtin_amdgcn_s_memrealtime();\n    __shared__ int ticket_old;\n")
rep("    const bool finish = nr == 1 || ticket_old == nr - 1;      // block-uniform\n","    const bool finish = nr == 1 || ticket_old == nr - 1;      // block-uniform\n    if (seg == 0) tl[6] = __builtin_amdgcn_s_memrealtime();\n")
rep("        if (s0 + l15 < N) {\n#pragma unroll\n            for (int q = 0; q < 24; ++q) {\n                const int qq = 24 * wid + q","        if (seg == 0) tl[7] = __builtin_amdgcn_s_memrealtime();\n        if (s0 + l15 < N) {\n#pragma unroll\n            for (int q = 0; q < 24; ++q) {\n                const int qq = 24 * wid + q")
rep("    }   // segment\n","    if (seg == 0) tl[3] = __builtin_amdgcn_s_memrealtime();\n    }   // segment\n")
# end of kernel: after the grid-level L1 reduction
k=s.index("// Temporal smoothness of the output joints, HuMoR's joints3d_smooth_loss")
e=s.rfind("}\n", 0, k)
s=s[:e]+"    if (threadIdx.x == 0 && blockIdx.x < 1024) {\n        tl[4] = __builtin_amdgcn_s_memrealtime();\n        for (int q = 0; q < 8; ++q) mesh_tl[blockIdx.x * 8 + q] = tl[q];\n    }\n}\n"+s[e+2:]
s+='''
extern "C" int32_t nemo_debug_mesh_timeline(unsigned long long* out) {
    return (int32_t)hipMemcpyFromSymbol(out, HIP_SYMBOL(mesh_tl), sizeof(unsigned long long) * 8 * 1024);
}
'''
open('smpl_tl.hip','w').write(s)
PY
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function -Wno-pass-failed -fno-slp-vectorize -c smpl_tl.hip -o /tmp/smpl_tl.o 2>&1 | grep -E "error" -A5 | head -20; /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC gemm.o pose.o /tmp/smpl_tl.o prior.o -o ../libnemo_hip_tl.so; rm -f smpl_tl.hip; ls -la ../libnemo_hip_tl.so
