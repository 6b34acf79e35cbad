#!/bin/bash
# Round 6.  Everything profiles/r06_kernel_trace_*.md, r06_pmc_traffic.md, r06_pmc_mfma.md and profiles/traffic.json are built from, in one
# gpurun call:   gpurun --timeout 2400 -- 'bash tools/profile_r06.sh'   then (here)   python3 tools/write_profiles_r06.py
# (rocprofv3 gets the program itself after `--`; counter passes are separate runs with --kernel-trace only.)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp && cd "$R"
O=gpurun_out/r06p
rm -rf $O; mkdir -p $O
B="--no-cpu-baseline --no-torch-gpu-baseline --no-extra-legs"
X="--repeat 1 --minibatch-steps 0"
C3="--instances 40 --dtype bf16"
# 1. kernel traces (graph replays): headline C2 (f32_split default), C2 with the fp32-MFMA blend, one-instance shard, C3 bf16
rocprofv3 --kernel-trace --stats -d $O/trace_c2 -o t -- python3 bench.py --steps 20 --warmup 2 $X $B > $O/trace_c2.log 2>&1
NEMO_MESH_BLEND=f32 NEMO_MLP_GEMM=f32 rocprofv3 --kernel-trace --stats -d $O/trace_c2f -o t -- python3 bench.py --steps 20 --warmup 2 $X $B > $O/trace_c2f.log 2>&1
rocprofv3 --kernel-trace --stats -d $O/trace_v1 -o t -- python3 bench.py --instances 1 --steps 20 --warmup 2 $X $B > $O/trace_v1.log 2>&1
rocprofv3 --kernel-trace --stats -d $O/trace_c3b -o t -- python3 bench.py $C3 --steps 10 --warmup 2 $X $B > $O/trace_c3b.log 2>&1
for c in c2 c2f v1 c3b; do
  python3 tools/prof_summary.py $O/trace_$c/t_results.db 35 > $O/summary_$c.md 2>&1
  python3 tools/step_timeline.py $O/trace_$c/t_results.db 12 > $O/timeline_$c.txt 2>&1
done
# 1b. the mesh kernel's LDS / L1 counters on the default (random-permutation) and the spatially structured body model
for bm in default locality; do
  NEMO_BENCH_LOCALITY=$([ $bm = locality ] && echo 1 || echo 0) NEMO_GRAPHS=0 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS TCP_TCC_READ_REQ_sum --output-format csv -d $O/pmc_lds_$bm -o l -- python3 bench.py --steps 3 --warmup 1 $X $B > $O/pmc_lds_$bm.log 2>&1
done
NEMO_BENCH_LOCALITY=1 rocprofv3 --kernel-trace --stats -d $O/trace_loc -o t -- python3 bench.py --steps 20 --warmup 2 $X $B > $O/trace_loc.log 2>&1
python3 tools/prof_summary.py $O/trace_loc/t_results.db 12 > $O/summary_loc.md 2>&1
# 2. HBM-side traffic, separate passes, eager launches (every kernel its own dispatch): C2 and C3 bf16
export NEMO_GRAPHS=0
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/pmc_fetch -o f -- python3 bench.py --steps 4 --warmup 1 $X $B > $O/pmc_f.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/pmc_write -o w -- python3 bench.py --steps 4 --warmup 1 $X $B > $O/pmc_w.log 2>&1
python3 tools/pmc_summary.py $O/pmc_fetch/f_results.db $O/pmc_write/w_results.db > $O/pmc_traffic.md 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/pmc_fetch_c3b -o f -- python3 bench.py $C3 --steps 3 --warmup 1 $X $B > $O/pmc_f_c3b.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/pmc_write_c3b -o w -- python3 bench.py $C3 --steps 3 --warmup 1 $X $B > $O/pmc_w_c3b.log 2>&1
python3 tools/pmc_summary.py $O/pmc_fetch_c3b/f_results.db $O/pmc_write_c3b/w_results.db > $O/pmc_traffic_c3b.md 2>&1
# 3. MFMA pipe counters (C2, C3 bf16)
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_VALU_MFMA_MOPS_BF16 --output-format csv -d $O/pmc_mfma -o m -- python3 bench.py --steps 4 --warmup 1 $X $B > $O/pmc_m.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_VALU_MFMA_MOPS_BF16 --output-format csv -d $O/pmc_mfma_c3b -o m -- python3 bench.py $C3 --steps 3 --warmup 1 $X $B > $O/pmc_m_c3b.log 2>&1
unset NEMO_GRAPHS
# 4. bench lines: bf16 at the headline size, C3 in fp32, the shard sizes, the sharded code path in a group of one
python3 bench.py --dtype bf16 --steps 30 --warmup 5 $B > $O/bench_c2_bf16.json 2>/dev/null
python3 bench.py --instances 40 --steps 20 --warmup 3 $B > $O/bench_c3_f32.json 2>/dev/null
python3 bench.py $C3 --steps 20 --warmup 3 $B > $O/bench_c3_bf16.json 2>/dev/null
for v in 1 2 4; do python3 bench.py --instances $v --steps 100 --warmup 5 $B > $O/bench_shard_v$v.json 2>/dev/null; done
for v in 1 8; do NEMO_BENCH_SHARD_OF_ONE=1 python3 bench.py --instances $v --steps 100 --warmup 5 $B 2>/dev/null | grep '^{' > $O/bench_group1_v$v.json; done
find $O -name "*.db" -delete
find $O -name "*.csv" -size +20M -delete
du -sh $O
