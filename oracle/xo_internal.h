// oracle/xo_internal.h -- shared internals of the oracle. TEST INFRASTRUCTURE ONLY.
#ifndef XO_INTERNAL_H
#define XO_INTERNAL_H
#include <complex>
#include <cstddef>
#define XO_EQUAL_ACCURACY 1e-6 /* XMIPP_EQUAL_ACCURACY (xmippCore xmipp_macros.h) */
#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif
namespace xo {
void c2c(const std::complex<double> *in, int n, int sign, std::complex<double> *out);
void prefilter2d(const double *in, int ydim, int xdim, double *coef);
void prefilter3d(double *c, int zdim, int ydim, int xdim);
double interp2d(const double *coef, int ydim, int xdim, int starty, int startx, double x, double y);
double realWRAP(double x, double x0, double xF);
void apply_geometry2d(int degree, const double *V1, int ydim, int xdim, const double *At, bool inv,
                      bool wrap, double *V2);
void rotation2DMatrix(double ang_deg, double *A);
void correlation_matrix(const double *m1, const double *m2, int ydim, int xdim, double *R);
double best_shift_mcorr(double *Mcorr, int ydim, int xdim, int maxShift, double &shiftX,
                        double &shiftY);
double correlation_index(const double *x, const double *y, size_t N);
inline int first_xmipp_index(int n) { return -(n / 2); }
inline int last_xmipp_index(int n) { return first_xmipp_index(n) + n - 1; }
}  // namespace xo
#endif
