// xmipp_movie_alignment_correlation -- same main as the reference's
// applications/programs/movie_alignment_correlation/movie_alignment_correlation_main.cpp (global alignment on the device)
#include "movie_programs.h"
int main(int argc, char **argv)
{
    mc::ProgMovieAlignmentCorrelation program;
    program.read(argc, argv);
    return program.tryRun();
}
