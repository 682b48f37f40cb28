#!/bin/bash
# Kernel tables (rocprofv3 --kernel-trace --stats) of the match-only step on the compact phantom, on the default gallery and with every
# data-dependent shortcut off: where an unfriendly map's time goes.   bash tools/diag_unfriendly.sh  -> gpurun_out/diag_match_*
root=$GRAFT_REPO_ROOT
cd /tmp; export TMPDIR=/tmp
for w in compact phantom; do
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$w -o $w -- python3 $root/bench.py --mode match --refs $w --steps 4 --warmup 2 --no-cpu-baseline --no-extra-legs > $root/gpurun_out/diag_match_$w.json 2> /tmp/err_$w.txt
cp /tmp/prof_$w/*kernel_stats.csv $root/gpurun_out/diag_match_${w}_kernel_stats.csv 2>/dev/null || find /tmp/prof_$w -name "*kernel_stats.csv" -exec cp {} $root/gpurun_out/diag_match_${w}_kernel_stats.csv \;
done
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_np -o np -- python3 $root/bench.py --mode match --no-prune --k0 1048576 --steps 4 --warmup 2 --no-cpu-baseline --no-extra-legs > $root/gpurun_out/diag_match_noprune.json 2> /tmp/err_np.txt
find /tmp/prof_np -name "*kernel_stats.csv" -exec cp {} $root/gpurun_out/diag_match_noprune_kernel_stats.csv \;
for f in /tmp/err_*.txt; do tail -n 3 $f; done
