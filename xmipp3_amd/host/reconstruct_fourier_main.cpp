// xmipp_reconstruct_fourier -- same main as the reference's applications/programs/reconstruct_fourier/reconstruct_fourier_main.cpp:
// ProgRecFourier (reconstruction/reconstruct_fourier.cpp), the double-precision program, on the device (xh_rf2_*)
#include "programs.h"
int main(int argc, char **argv)
{
    mc::ProgRecFourierAccel program;
    program.rfArithmetic = true;
    program.read(argc, argv);
    return program.tryRun();
}
