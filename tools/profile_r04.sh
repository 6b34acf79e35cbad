#!/bin/bash
# Everything profiles/r04_* is built from, in one gpurun call:   gpurun --timeout 3000 -- 'bash tools/profile_r04.sh'
# (rocprofv3 gets the program itself after `--`; counter passes are separate runs with --kernel-trace only.)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp && cd "$R"
O=gpurun_out/r04p
rm -rf $O; mkdir -p $O
B="--no-cpu-baseline --no-torch-gpu-baseline --no-extra-legs"
X="--repeat 1 --minibatch-steps 0"
# 1. kernel traces (graph replays): headline C2, one-instance shard, C3 bf16
rocprofv3 --kernel-trace --stats -d $O/trace_c2 -o t -- python3 bench.py --steps 20 --warmup 2 $X $B > $O/trace_c2.log 2>&1
rocprofv3 --kernel-trace --stats -d $O/trace_v1 -o t -- python3 bench.py --instances 1 --steps 20 --warmup 2 $X $B > $O/trace_v1.log 2>&1
rocprofv3 --kernel-trace --stats -d $O/trace_c3b -o t -- python3 bench.py --instances 40 --dtype bf16 --steps 10 --warmup 2 $X $B > $O/trace_c3b.log 2>&1
# the fit phases: warm-up, camera fit, minibatch steps (tools/bench_phases.py)
rocprofv3 --kernel-trace --stats -d $O/trace_phases -o t -- python3 tools/bench_phases.py > $O/trace_phases.log 2>&1
for c in c2 v1 c3b phases; do
  python3 tools/prof_summary.py $O/trace_$c/t_results.db 35 > $O/summary_$c.md 2>&1
done
for c in c2 v1 c3b; do
  python3 tools/step_timeline.py $O/trace_$c/t_results.db 12 > $O/timeline_$c.txt 2>&1
done
# 2. HBM-side traffic, separate passes, eager launches (every kernel its own dispatch)
export NEMO_GRAPHS=0
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/pmc_fetch -o f -- python3 bench.py --steps 4 --warmup 1 $X $B > $O/pmc_f.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/pmc_write -o w -- python3 bench.py --steps 4 --warmup 1 $X $B > $O/pmc_w.log 2>&1
python3 tools/pmc_summary.py $O/pmc_fetch/f_results.db $O/pmc_write/w_results.db > $O/pmc_traffic.md 2>&1
# 3. MFMA pipe counters
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F32 --output-format csv -d $O/pmc_mfma -o m -- python3 bench.py --steps 4 --warmup 1 $X $B > $O/pmc_m.log 2>&1
unset NEMO_GRAPHS
# 4. bench lines: the full default line (with every leg), the shard sizes, bf16 at the headline size, the sharded code path in a group of one
python3 bench.py > $O/bench_c2_full.json 2> $O/bench_c2_full.err
python3 bench.py --dtype bf16 --steps 30 --warmup 5 $B > $O/bench_c2_bf16.json 2>/dev/null
python3 bench.py --instances 40 --steps 20 --warmup 3 $B > $O/bench_c3_f32.json 2>/dev/null
for v in 1 2 4; do python3 bench.py --instances $v --steps 100 --warmup 5 $B > $O/bench_shard_v$v.json 2>/dev/null; done
for v in 1 8; do NEMO_BENCH_SHARD_OF_ONE=1 python3 bench.py --instances $v --steps 100 --warmup 5 $B 2>/dev/null | grep '^{' > $O/bench_group1_v$v.json; done
# 5. un-profiled contribution of single kernels to the step (tools/ablate.py) at the headline size
bash tools/ablate.sh 8 nemo_v2v_fused nemo_gemm_f32@2400x207x20670 nemo_gemm_f32@2401x1000x1000 nemo_gemm_f32@1000x1000x2401 nemo_kp_bwd_ex nemo_kp_fwd nemo_gmm_fwd_bwd nemo_fk_fwd nemo_fk_bwd nemo_phase_embed_bwd_colsum nemo_adam_step_dev_if > $O/ablate_v8.txt 2>&1
# 6. harnesses
./tools/gemm_glds_dev adj 2400 > $O/adj_2400.txt 2>&1
./tools/gemm_glds_dev adj 1200 | tail -7 > $O/adj_1200.txt 2>&1
./tools/gemm_glds_dev adj 8192 | tail -7 > $O/adj_8192.txt 2>&1
find $O -name "*.db" -size +30M -delete
du -sh $O; tail -c 300 $O/bench_c2_full.json
