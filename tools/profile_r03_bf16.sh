#!/bin/bash
# bf16 lines + C3 bf16 kernel trace of round 3 (after the bf16-in-memory chain): gpurun -- 'bash tools/profile_r03_bf16.sh'
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp && cd "$R"
O=gpurun_out/r03p
mkdir -p $O
B="--no-cpu-baseline --no-torch-gpu-baseline"
python3 bench.py --instances 40 --dtype bf16 --steps 20 --warmup 3 $B > $O/bench_c3_bf16.json 2>/dev/null
NEMO_MESH_SPLIT=0 python3 bench.py --instances 40 --dtype bf16 --steps 20 --warmup 3 $B > $O/bench_c3_bf16_nosplit.json 2>/dev/null
python3 bench.py --dtype bf16 --steps 30 --warmup 5 $B > $O/bench_c2_bf16.json 2>/dev/null
NEMO_BF16_MEM=0 python3 bench.py --instances 40 --dtype bf16 --steps 20 --warmup 3 $B > $O/bench_c3_bf16_onthefly.json 2>/dev/null
rocprofv3 --kernel-trace --stats -d $O/trace_c3b -o t -- python3 bench.py --instances 40 --dtype bf16 --steps 10 --warmup 2 --repeat 1 --minibatch-steps 0 $B > $O/trace_c3b.log 2>&1
python3 tools/prof_summary.py $O/trace_c3b/t_results.db 25 > $O/summary_c3b.md 2>&1
python3 tools/step_timeline.py $O/trace_c3b/t_results.db 12 > $O/timeline_c3b.txt 2>&1
python3 tools/bench_bf16mem.py > $O/bf16mem_gemm.txt 2>&1
find $O -name "*.db" -size +30M -delete
