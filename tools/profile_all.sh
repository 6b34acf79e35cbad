#!/bin/bash
# One gpurun call that collects everything tools/write_profiles.py turns into profiles/r01b_*, r01c_*:
#   gpurun --timeout 1500 -- 'bash tools/profile_all.sh'
# (rocprofv3 gets the program itself after `--`; the counter passes are separate runs with --kernel-trace only.)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp && cd "$R"
O=gpurun_out
rm -rf $O/prof_trace $O/pmc_fetch $O/pmc_write
mkdir -p $O
rocprofv3 --kernel-trace --stats -d $O/prof_trace -o r01b -- python3 bench.py --steps 20 --warmup 2 --no-cpu-baseline --no-torch-gpu-baseline > $O/prof_b.log 2>&1
export NEMO_GRAPHS=0
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/pmc_fetch -o f -- python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-torch-gpu-baseline > $O/pmc_f.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/pmc_write -o w -- python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-torch-gpu-baseline > $O/pmc_w.log 2>&1
unset NEMO_GRAPHS
if [ "${1:-}" != "quick" ]; then
  python3 tools/bench_gemm.py mlp,rot_out,vposer,mq,dpf,pose_blend_bwd --sweep 20 > $O/gemm_sweep_2400.txt 2>&1
  python3 tools/bench_gemm.py mlp,rot_out,vposer 20 300 --sweep > $O/gemm_sweep_300.txt 2>&1
  [ -x tools/mfma_peak ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 tools/mfma_peak.hip -o tools/mfma_peak
  ./tools/mfma_peak 2000 > $O/mfma_peak.txt 2>&1
  python3 tools/bench_torch_mm.py > $O/torch_mm.txt 2>&1
fi
python3 bench.py > $O/bench_full.json 2> $O/bench_full.err
tail -c 600 $O/bench_full.json
ls $O/prof_trace $O/pmc_fetch $O/pmc_write
