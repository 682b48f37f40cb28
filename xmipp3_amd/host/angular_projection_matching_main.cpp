// xmipp_angular_projection_matching -- same main as the reference's
// applications/programs/angular_projection_matching/angular_projection_matching_main.cpp
#include "programs.h"
int main(int argc, char **argv)
{
    mc::ProgAngularProjectionMatching program;
    program.read(argc, argv);
    return program.tryRun();
}
