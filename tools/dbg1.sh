cd $GRAFT_REPO_ROOT
timeout 600 python3 -m pytest tests/test_gpu_pm.py -q -x -k "branch_and_bound or full_size or exact_indices or ties" 2>&1 | tail -3
bash tools/trace_onestream.sh r05d 2>/dev/null | grep -i "idft_max3\|span"
