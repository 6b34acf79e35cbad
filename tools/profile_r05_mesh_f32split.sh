#!/bin/bash
# what bounds mesh_v2v_fused_kernel<3, true> (the bf16 mesh kernel of C3, 40 x 300)?  Instruction-mix and
# stall counters in separate --pmc passes (8 SQ slots each; eager launches so that every kernel is its own dispatch), summarised
# per kernel by tools/pmc_summary_kernel.py.   gpurun -- 'bash tools/profile_r05_mesh_b16.sh'
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp && cd "$R"
O=${OUT:-gpurun_out/r05/pmc_mesh_f32s}
rm -rf $O; mkdir -p $O
export NEMO_GRAPHS=0
B="--steps 3 --warmup 1 --repeat 1 --minibatch-steps 0 --no-cpu-baseline --no-torch-gpu-baseline"
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM"
P2="SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_MFMA"
P3="SQ_WAVE_CYCLES SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_WAVES"
P4="SQ_WAVE_CYCLES SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_INT32 SQ_LDS_UNALIGNED_STALL SQ_LDS_DATA_FIFO_FULL"
i=0
for P in "$P1" "$P2" "$P3" "$P4"; do
    i=$((i + 1))
    rocprofv3 --kernel-trace --pmc $P GRBM_GUI_ACTIVE --output-format csv -d $O/p$i -o c -- python3 bench.py $B > $O/p$i.log 2>&1
    f=$(find $O/p$i -name "*counter_collection.csv" | head -1)
    echo "## pass $i: $P" >> $O/summary.txt
    if [ -n "$f" ]; then python3 tools/pmc_summary_kernel.py "$f" mesh_v2v_fused >> $O/summary.txt 2>&1; else echo "(no counter file; see p$i.log)" >> $O/summary.txt; tail -3 $O/p$i.log >> $O/summary.txt; fi
done
find $O -name "*.csv" -size +20M -delete
cat $O/summary.txt
