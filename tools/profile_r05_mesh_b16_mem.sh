#!/bin/bash
# VERDICT r04 item 2, memory side: L1 (TCP) / L2 (TCC) request counters of the bf16 mesh kernel (C3, 40 x 300), separate --pmc
# passes, eager launches.   gpurun -- 'bash tools/profile_r05_mesh_b16_mem.sh'
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp && cd "$R"
O=${OUT:-gpurun_out/r05/pmc_mesh_b16_mem}
rm -rf $O; mkdir -p $O
export NEMO_GRAPHS=0
rocprofv3 -L > $O/counters.txt 2>&1
B="--instances 40 --dtype bf16 --steps 3 --warmup 1 --repeat 1 --minibatch-steps 0 --no-cpu-baseline --no-torch-gpu-baseline"
P1="TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum"
P2="TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_READ_sum"
P3="TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_TA_TCP_STATE_READ_sum TCP_TD_TCP_STALL_CYCLES_sum"
P4="TA_TA_BUSY_sum TA_BUFFER_LOAD_WAVEFRONTS_sum TA_DATA_STALL_BY_TC_CYCLES_sum TD_TD_BUSY_sum"
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
