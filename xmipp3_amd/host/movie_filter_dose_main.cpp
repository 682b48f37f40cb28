// xmipp_movie_filter_dose -- same main as the reference's applications/programs/movie_filter_dose/movie_filter_dose_main.cpp
#include "movie_programs.h"
int main(int argc, char **argv)
{
    mc::ProgMovieFilterDose program;
    program.read(argc, argv);
    return program.tryRun();
}
