// xmipp_resolution_fsc -- same main as the reference's
// applications/programs/resolution_fsc/resolution_fsc_main.cpp
#include "programs.h"
int main(int argc, char **argv)
{
    mc::ProgResolutionFsc program;
    program.read(argc, argv);
    return program.tryRun();
}
