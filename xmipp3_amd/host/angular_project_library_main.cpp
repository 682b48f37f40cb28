// xmipp_angular_project_library -- same main as the reference's
// applications/programs/angular_project_library/angular_project_library_main.cpp
#include "programs.h"
int main(int argc, char **argv)
{
    mc::ProgAngularProjectLibrary program;
    program.read(argc, argv);
    return program.tryRun();
}
