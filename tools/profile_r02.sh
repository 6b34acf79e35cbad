#!/bin/bash
# Everything profiles/r02_* is built from, in one gpurun call:   gpurun --timeout 2400 -- 'bash tools/profile_r02.sh'
# (rocprofv3 gets the program itself after `--`; counter passes are separate runs with --kernel-trace only.)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp && cd "$R"
O=gpurun_out/r02p
rm -rf $O; mkdir -p $O
B="--no-cpu-baseline --no-torch-gpu-baseline"
# 1. kernel traces (graph replays): headline C2, one-instance shard, C3 in both arithmetics
rocprofv3 --kernel-trace --stats -d $O/trace_c2 -o t -- python3 bench.py --steps 20 --warmup 2 $B > $O/trace_c2.log 2>&1
rocprofv3 --kernel-trace --stats -d $O/trace_v1 -o t -- python3 bench.py --instances 1 --steps 20 --warmup 2 $B > $O/trace_v1.log 2>&1
rocprofv3 --kernel-trace --stats -d $O/trace_c3 -o t -- python3 bench.py --instances 40 --steps 10 --warmup 2 $B > $O/trace_c3.log 2>&1
rocprofv3 --kernel-trace --stats -d $O/trace_c3b -o t -- python3 bench.py --instances 40 --dtype bf16 --steps 10 --warmup 2 $B > $O/trace_c3b.log 2>&1
for c in c2 v1 c3 c3b; do
  st=35; [ $c = c3 ] && st=25; [ $c = c3b ] && st=25
  python3 tools/prof_summary.py $O/trace_$c/t_results.db $st > $O/summary_$c.md 2>&1
  python3 tools/step_timeline.py $O/trace_$c/t_results.db 12 > $O/timeline_$c.txt 2>&1
done
# 2. HBM-side traffic, separate passes, eager launches (every kernel its own dispatch)
export NEMO_GRAPHS=0
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/pmc_fetch -o f -- python3 bench.py --steps 4 --warmup 1 $B > $O/pmc_f.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/pmc_write -o w -- python3 bench.py --steps 4 --warmup 1 $B > $O/pmc_w.log 2>&1
python3 tools/pmc_summary.py $O/pmc_fetch/f_results.db $O/pmc_write/w_results.db > $O/pmc_traffic.md 2>&1
# 3. MFMA pipe / LDS counters
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F32 --output-format csv -d $O/pmc_mfma -o m -- python3 bench.py --steps 4 --warmup 1 $B > $O/pmc_m.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES --output-format csv -d $O/pmc_lds -o l -- python3 bench.py --steps 4 --warmup 1 $B > $O/pmc_l.log 2>&1
unset NEMO_GRAPHS
# 4. bench lines of every BASELINE configuration + the shard sizes + the full default line
python3 bench.py > $O/bench_c2_full.json 2> $O/bench_c2_full.err
python3 bench.py --instances 40 --steps 20 --warmup 3 $B > $O/bench_c3_f32.json 2>/dev/null
python3 bench.py --instances 40 --dtype bf16 --steps 20 --warmup 3 $B > $O/bench_c3_bf16.json 2>/dev/null
python3 bench.py --dtype bf16 --steps 30 --warmup 5 $B > $O/bench_c2_bf16.json 2>/dev/null
python3 bench.py --instances 256 --frames 1024 --steps 5 --warmup 2 $B > $O/bench_c4.json 2>/dev/null
for v in 1 2 4; do python3 bench.py --instances $v --steps 100 --warmup 5 $B > $O/bench_shard_v$v.json 2>/dev/null; done
# 5. GEMM harness
timeout 300 ./tools/gemm_glds_dev calib > $O/gemm_calib.txt 2>&1
timeout 300 ./tools/gemm_glds_dev time 2400 > $O/gemm_time_2400.txt 2>&1
timeout 300 ./tools/gemm_glds_dev time 300 > $O/gemm_time_300.txt 2>&1
find $O -name "*.db" -size +30M -delete
du -sh $O; tail -c 400 $O/bench_c2_full.json
