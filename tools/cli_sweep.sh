#!/bin/bash
# A/B of the host side of the two programs on one box: reader lanes x copy engines (tools/bench_cli.py --no-library --keep reuses nothing:
# each line regenerates the files, ~3 s).  Usage: tools/cli_sweep.sh [box] [rows] [unique]
cd "$(dirname "$0")/.."
BOX=${1:-256}; ROWS=${2:-65536}; UNIQ=${3:-16384}
mkdir -p gpurun_out
for rd in ${READERS:-16 32 64}; do
  for sdma in ${SDMA:-unset 1}; do
    envarg="${EXTRA_ENV}"; [ "$sdma" != unset ] && envarg="$envarg --env HSA_ENABLE_SDMA=$sdma"
    python tools/bench_cli.py --box $BOX --particles $ROWS --unique $UNIQ --readers $rd --no-library $envarg 2>/dev/null | python -c "
import json,sys
d=json.load(sys.stdin)
for p in ('xmipp_angular_projection_matching','xmipp_reconstruct_fourier_accel'):
    t=d[p]['timing_s']; print('readers $rd sdma $sdma', p[6:28], 'loop/s %.0f'%t['images_per_s_loop'], 'stall %.3f device %.3f load %.3f h2dwait %.3f setup %.3f total %.3f'%(t['stall'],t['device'],t['load'],t['h2d_wait'],t['setup'],t['total']))
"
  done
done
