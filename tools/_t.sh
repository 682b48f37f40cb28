root=$GRAFT_REPO_ROOT
cd $root
python3 -m pytest tests/test_gpu_flexalign.py -x -q 2>&1 | tail -3
cd /tmp; export TMPDIR=/tmp
for t in base b3off warp0 base b3off; do
lib=$root/xmipp3_amd/libxmipp_hip_$t.so; [ "$t" == "base" ] && lib=$root/xmipp3_amd/libxmipp_hip.so
export XMIPP_HIP_LIB=$lib
rm -rf /tmp/prof_$t
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$t -o x -- python3 $root/bench.py --mode flexalign --steps 3 --warmup 1 --no-cpu-baseline --no-extra-legs --fa-lanes 1 > /tmp/out_$t.json 2>/dev/null
f=$(find /tmp/prof_$t -name '*kernel_stats.csv' | head -1)
echo "== $t"; grep "k_fa_warp" $f | python3 -c "
import sys,csv
for r in csv.reader(sys.stdin): print('   ', r[0][:50], 'calls', r[1], 'avg ms %.4f'%(float(r[3])/1e6))"
done
