cd $GRAFT_REPO_ROOT
timeout 1100 python3 -m pytest tests -q -m gpu > gpurun_out/r05a_tests3.txt 2>&1; tail -12 gpurun_out/r05a_tests3.txt | cut -c1-300
