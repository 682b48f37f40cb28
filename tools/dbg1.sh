cd $GRAFT_REPO_ROOT
timeout 600 python3 -m pytest tests/test_gpu_rf.py tests/test_cli.py -q -x -k "shift or second_iteration or pipeline" 2>&1 | tail -3
bash tools/trace_onestream.sh r05e 2>/dev/null | grep -i "k_rf_shift\|span"
