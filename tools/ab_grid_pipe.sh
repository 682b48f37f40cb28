for r in 1 2 3; do
  for t in base pipe2 pipe4 halves4; do
    lib=$PWD/xmipp3_amd/libxmipp_hip_$t.so; [ "$t" == "base" ] && lib=$PWD/xmipp3_amd/libxmipp_hip.so
    for w in 12 16; do
      echo -n "$t waves $w  "; XMIPP_HIP_LIB=$lib python3 tools/bench_grid.py --opt grid_waves=$w 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_launch'])"
    done
  done
done
