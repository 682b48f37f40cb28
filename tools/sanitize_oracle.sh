#!/bin/bash
# The CPU suite with the oracle built under AddressSanitizer + UndefinedBehaviorSanitizer (the GPU box has no sanitizer for device
# code; the CPU restatement every parity test leans on gets one here).   bash tools/sanitize_oracle.sh
set -e
root=$(cd "$(dirname "$0")/.." && pwd)
cd $root/oracle
g++ -O1 -g -fsanitize=address,undefined -fno-omit-frame-pointer -ffp-contract=off -fopenmp -fPIC -std=c++17 -shared -o /tmp/liboracle_asan.so xo_*.cpp
cd $root
ASAN=$(gcc -print-file-name=libasan.so); UBSAN=$(gcc -print-file-name=libubsan.so)
ASAN_OPTIONS=detect_leaks=0:verify_asan_link_order=0 LD_PRELOAD="$ASAN $UBSAN" XO_ORACLE_LIB=/tmp/liboracle_asan.so OMP_NUM_THREADS=4 \
  python -m pytest tests -q -m "not gpu" --deselect tests/test_distributed.py
