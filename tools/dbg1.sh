cd /tmp && export TMPDIR=/tmp
root=$GRAFT_REPO_ROOT
for f in 1 0; do
rm -rf /tmp/pfa; export XH_PREFILTER_FORM=$f
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pfa -o fa -- python3 $root/bench.py --mode flexalign --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
c=$(find /tmp/pfa -name '*kernel_stats.csv' | head -1)
cp $c $root/gpurun_out/r05b_fa_stats_form$f.csv
python3 $root/tools/kstats.py $c 14 5 | cut -c1-150
done
