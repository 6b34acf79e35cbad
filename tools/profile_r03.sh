#!/bin/bash
# Everything profiles/r03_* is built from, in one gpurun call:   gpurun --timeout 3000 -- 'bash tools/profile_r03.sh'
# (rocprofv3 gets the program itself after `--`; counter passes are separate runs with --kernel-trace only.)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp && cd "$R"
O=gpurun_out/r03p
rm -rf $O; mkdir -p $O
B="--no-cpu-baseline --no-torch-gpu-baseline"
# 1. kernel traces (graph replays): headline C2, one- and two-instance shards
rocprofv3 --kernel-trace --stats -d $O/trace_c2 -o t -- python3 bench.py --steps 20 --warmup 2 --repeat 1 --minibatch-steps 0 $B > $O/trace_c2.log 2>&1
rocprofv3 --kernel-trace --stats -d $O/trace_v1 -o t -- python3 bench.py --instances 1 --steps 20 --warmup 2 --repeat 1 --minibatch-steps 0 $B > $O/trace_v1.log 2>&1
rocprofv3 --kernel-trace --stats -d $O/trace_v2 -o t -- python3 bench.py --instances 2 --steps 20 --warmup 2 --repeat 1 --minibatch-steps 0 $B > $O/trace_v2.log 2>&1
for c in c2 v1 v2; do
  python3 tools/prof_summary.py $O/trace_$c/t_results.db 35 > $O/summary_$c.md 2>&1
  python3 tools/step_timeline.py $O/trace_$c/t_results.db 12 > $O/timeline_$c.txt 2>&1
done
# 2. HBM-side traffic, separate passes, eager launches (every kernel its own dispatch)
export NEMO_GRAPHS=0
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/pmc_fetch -o f -- python3 bench.py --steps 4 --warmup 1 --repeat 1 --minibatch-steps 0 $B > $O/pmc_f.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/pmc_write -o w -- python3 bench.py --steps 4 --warmup 1 --repeat 1 --minibatch-steps 0 $B > $O/pmc_w.log 2>&1
python3 tools/pmc_summary.py $O/pmc_fetch/f_results.db $O/pmc_write/w_results.db > $O/pmc_traffic.md 2>&1
# 3. MFMA pipe counters
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F32 --output-format csv -d $O/pmc_mfma -o m -- python3 bench.py --steps 4 --warmup 1 --repeat 1 --minibatch-steps 0 $B > $O/pmc_m.log 2>&1
unset NEMO_GRAPHS
# 4. bench lines: the full default line, every BASELINE configuration, the shard sizes, the sharded code path in a group of one
python3 bench.py > $O/bench_c2_full.json 2> $O/bench_c2_full.err
python3 bench.py --instances 40 --steps 20 --warmup 3 $B > $O/bench_c3_f32.json 2>/dev/null
python3 bench.py --instances 40 --dtype bf16 --steps 20 --warmup 3 $B > $O/bench_c3_bf16.json 2>/dev/null
python3 bench.py --dtype bf16 --steps 30 --warmup 5 $B > $O/bench_c2_bf16.json 2>/dev/null
python3 bench.py --instances 256 --frames 1024 --steps 5 --warmup 2 --repeat 3 --minibatch-steps 20 $B > $O/bench_c4.json 2>/dev/null
for v in 1 2 4; do python3 bench.py --instances $v --steps 100 --warmup 5 $B > $O/bench_shard_v$v.json 2>/dev/null; done
for v in 1 2 4; do NEMO_BENCH_SHARD_OF_ONE=1 python3 bench.py --instances $v --steps 100 --warmup 5 $B 2>/dev/null | grep '^{' > $O/bench_group1_v$v.json; done
# 5. un-profiled contribution of single kernels to the step (NEMO_ABLATE) at the shard sizes and the headline size
bash tools/ablate_r03_final.sh > /dev/null 2>&1
find $O -name "*.db" -size +30M -delete
du -sh $O; tail -c 300 $O/bench_c2_full.json
