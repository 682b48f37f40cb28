// xmipp_ctf_correct_wiener2d -- same main as the reference's
// applications/programs/ctf_correct_wiener2d/ctf_correct_wiener2d_main.cpp
#include "ctf_programs.h"
int main(int argc, char **argv)
{
    mc::ProgCorrectWiener2D program;
    program.read(argc, argv);
    return program.tryRun();
}
