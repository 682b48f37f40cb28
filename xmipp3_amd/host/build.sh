#!/bin/bash
# Builds the host programs against libxmipp_hip.so (plain g++: the host side has no HIP code).
set -e
cd "$(dirname "$0")"
mkdir -p ../bin
CXX=${CXX:-g++}
FLAGS="-O2 -std=c++17 -pthread -Wall -Wno-unused-function"
LINK="-L.. -lxmipp_hip -Wl,-rpath,\$ORIGIN/.. -Wl,-rpath,/opt/rocm/lib"
$CXX $FLAGS angular_projection_matching_main.cpp -o ../bin/xmipp_angular_projection_matching $LINK &
$CXX $FLAGS reconstruct_fourier_accel_main.cpp -o ../bin/xmipp_reconstruct_fourier_accel $LINK &
$CXX $FLAGS reconstruct_fourier_main.cpp -o ../bin/xmipp_reconstruct_fourier $LINK &
$CXX $FLAGS angular_project_library_main.cpp -o ../bin/xmipp_angular_project_library $LINK &
$CXX $FLAGS resolution_fsc_main.cpp -o ../bin/xmipp_resolution_fsc $LINK &
$CXX $FLAGS ctf_phase_flip_main.cpp -o ../bin/xmipp_ctf_phase_flip $LINK &
$CXX $FLAGS ctf_correct_wiener2d_main.cpp -o ../bin/xmipp_ctf_correct_wiener2d $LINK &
$CXX $FLAGS movie_alignment_correlation_main.cpp -o ../bin/xmipp_movie_alignment_correlation $LINK &
$CXX $FLAGS movie_filter_dose_main.cpp -o ../bin/xmipp_movie_filter_dose $LINK &
wait
echo "built $(cd ../bin && pwd)/xmipp_{angular_projection_matching,reconstruct_fourier_accel,reconstruct_fourier,angular_project_library,resolution_fsc,ctf_phase_flip,ctf_correct_wiener2d,movie_alignment_correlation,movie_filter_dose}"
