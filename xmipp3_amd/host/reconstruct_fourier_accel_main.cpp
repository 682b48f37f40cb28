// xmipp_reconstruct_fourier_accel -- same main as the
// reference's applications/programs/reconstruct_fourier_accel/reconstruct_fourier_accel_main.cpp
#include "programs.h"
int main(int argc, char **argv)
{
    mc::ProgRecFourierAccel program;
    program.read(argc, argv);
    return program.tryRun();
}
