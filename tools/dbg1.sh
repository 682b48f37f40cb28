cd $GRAFT_REPO_ROOT
timeout 900 python3 -m pytest tests/test_gpu_pm.py -q -x -k "translat or s6 or shift" 2>&1 | tail -3
bash tools/trace_onestream.sh r05e 2>/dev/null | grep -i "bestshift\|span"
