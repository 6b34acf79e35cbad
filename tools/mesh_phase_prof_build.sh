#!/bin/bash
# builds nemo_cvpr2023_amd/libnemo_hip_abl.so = current smpl.hip + cycle-counter probes at the /*prof:..*/ markers of the
# fused mesh kernel
cd "$(dirname "$0")/../nemo_cvpr2023_amd/csrc" && python - <<'PY'
s=open('smpl.hip').read()
def rep(a,b):
    global s
    assert a in s, a[:50]
    s=s.replace(a,b,1)
RC='__builtin_readcyclecounter()'
rep("template <int MODE, bool SPARSE = false>\n__global__ __launch_bounds__(256, 2) void mesh_v2v_fused_kernel(","__device__ unsigned long long mesh_prof[8 * 1024];\ntemplate <int MODE, bool SPARSE = false>\n__global__ __launch_bounds__(256, 2) void mesh_v2v_fused_kernel(")
rep("/*prof:init*/","unsigned long long pr_blend = 0, pr_rec = 0, pr_orig = 0, pr_tiles = 0, pr_sk = 0, pr_va = 0, pr_ad = 0, pr_st = 0;")
rep("/*prof:c0*/","unsigned long long c0 = "+RC+";")
rep("/*prof:c1*/","unsigned long long c1 = "+RC+";")
rep("/*prof:c2*/","unsigned long long c2 = "+RC+";")
rep("/*prof:q0*/","unsigned long long q0 = "+RC+";")
rep("/*prof:q1*/","unsigned long long q1 = "+RC+";")
rep("/*prof:q2*/","unsigned long long q2 = "+RC+"; pr_sk += q1 - q0; pr_va += q2 - q1;")
rep("/*prof:q2e*/","pr_ad += "+RC+" - q2;")
rep("/*prof:q3*/","unsigned long long q3 = "+RC+";")
rep("/*prof:c3*/","unsigned long long c3 = "+RC+"; pr_blend += c1 - c0; pr_rec += c2 - c1; pr_orig += c3 - c2; pr_tiles += 1; pr_st += c3 - q3;")
rep("/*prof:out*/","if (threadIdx.x == 0 && blockIdx.x < 1024 && seg == 0) { unsigned long long* o = mesh_prof + blockIdx.x * 8; o[0] = pr_blend; o[1] = pr_rec; o[2] = pr_orig; o[3] = pr_tiles; o[4] = pr_sk; o[5] = pr_va; o[6] = pr_ad; o[7] = pr_st; }")
s+='''
extern "C" int32_t nemo_debug_mesh_prof(unsigned long long* out) {
    return (int32_t)hipMemcpyFromSymbol(out, HIP_SYMBOL(mesh_prof), sizeof(unsigned long long) * 8 * 1024);
}
'''
open('smpl_prof.hip','w').write(s)
PY
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-slp-vectorize -Wno-unused-function -Wno-pass-failed -c smpl_prof.hip -o /tmp/smpl_prof.o 2>&1 | grep -E "error" -A5 | head; /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC gemm.o pose.o /tmp/smpl_prof.o prior.o -o ../libnemo_hip_abl.so; rm -f smpl_prof.hip
