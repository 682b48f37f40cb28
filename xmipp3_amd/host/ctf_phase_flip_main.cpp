// xmipp_ctf_phase_flip -- same main as the reference's
// applications/programs/ctf_phase_flip/ctf_phase_flip_main.cpp
#include "ctf_programs.h"
int main(int argc, char **argv)
{
    mc::ProgCTFPhaseFlipping program;
    program.read(argc, argv);
    return program.tryRun();
}
