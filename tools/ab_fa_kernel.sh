#!/bin/bash
# Kernel-level A/B of FlexAlign variants (SRC=xh_flexalign tools/build_variant.sh): rocprofv3 kernel stats of one lane of bench.py --mode flexalign,
# the average duration of the kernels whose names match, alternating, two rounds.   bash tools/ab_fa_kernel.sh <kernel name regex> tagA tagB ...
pat=$1; shift
root=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp; export TMPDIR=/tmp
for r in 1 2; do
for t in "$@"; do
lib=$root/xmipp3_amd/libxmipp_hip_$t.so; [ "$t" == "base" ] && lib=$root/xmipp3_amd/libxmipp_hip.so
export XMIPP_HIP_LIB=$lib
rm -rf /tmp/prof_$t
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$t -o x -- python3 $root/bench.py --mode flexalign --steps 3 --warmup 1 --no-cpu-baseline --no-extra-legs --fa-lanes 1 > /dev/null 2>&1
f=$(find /tmp/prof_$t -name '*kernel_stats.csv' | head -1)
echo "== $t"; grep -E "$pat" $f | python3 -c "
import sys,csv
for r in csv.reader(sys.stdin): print('   ', r[0][:60], 'calls', r[1], 'avg ms %.4f'%(float(r[3])/1e6))"
done; done
