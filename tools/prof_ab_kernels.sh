#!/bin/bash
# per-kernel average durations of the headline step under two environments (kernel trace, graph replays):
#   gpurun -- 'bash tools/prof_ab_kernels.sh "NEMO_ORDERED_REDUCE=0" "NEMO_ORDERED_REDUCE=1"'
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp && cd "$R"
O=gpurun_out/r05/ab_kernels
rm -rf $O; mkdir -p $O
i=0
for kv in "$@"; do
    i=$((i + 1))
    env $kv rocprofv3 --kernel-trace --stats -d $O/t$i -o t -- python3 bench.py --no-cpu-baseline --no-torch-gpu-baseline --no-extra-legs --steps 10 --warmup 2 --repeat 1 ${EXTRA:-} > $O/t$i.log 2>&1
    echo "== $kv" >> $O/summary.txt
    python3 tools/prof_summary.py $O/t$i/t_results.db 1 2>/dev/null | awk -F'|' 'NR>2 && NF>4 {printf "%-70s calls %8s avg us %8s\n", substr($2,1,70), $3, $5}' | head -40 >> $O/summary.txt
done
find $O -name "*.db" -size +30M -delete
cat $O/summary.txt
