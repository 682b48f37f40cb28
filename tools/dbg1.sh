cd /tmp && export TMPDIR=/tmp
root=$GRAFT_REPO_ROOT
rm -rf /tmp/pfa
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pfa -o fa -- python3 $root/bench.py --mode flexalign --steps 2 --warmup 1 --no-cpu-baseline --no-extra-legs --fa-lanes 1 > $root/gpurun_out/r05_fa_under_rocprof.json 2> $root/gpurun_out/r05_fa_under_rocprof.err
echo "rc $?"
c=$(find /tmp/pfa -name '*kernel_stats.csv' | head -1)
cp $c $root/gpurun_out/r05_fa_kernel_stats.csv
python3 $root/tools/kstats.py $c 24 4 | cut -c1-170
