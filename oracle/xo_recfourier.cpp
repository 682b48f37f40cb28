// oracle/xo_recfourier.cpp -- Kaiser-Bessel tables, CTF evaluation and the
// ProgRecFourierAccel gridding/finaliser.  TEST INFRASTRUCTURE ONLY.
//
// Follows reconstruction/reconstruct_fourier_accel.cpp (RFA) line by line:
//   produceSideinfo  RFA:175-257      cropAndShift      RFA:271-298
//   preloadBuffer    RFA:300-388      preloadCTF        RFA:548-592
//   processVoxel     RFA:595-625      processVoxelBlob  RFA:627-700
//   processProjection RFA:710-763     geometry helpers  RFA:440-522
//   applyBlob        RFA:793-831      convertToExpectedSpace RFA:834-851
//   mirrorAndCrop    RFA:853-887      forceHermitianSymmetry RFA:889-906
//   processWeights   RFA:908-924      finishComputations RFA:1002-1055
// data/blobs.cpp:37-92,144-172 (kaiser_*), data/ctf.h:452-502,1002-1029,
// data/ctf.cpp:645-679,1392-1402.  Bessel functions are the Numerical-Recipes
// forms xmippCore ships (in-tree float copies:
// reconstruction_cuda/cuda_gpu_reconstruct_fourier.cpp:85-148).
//
// Quirk decisions: the CPU program computes the CTF arrays but
// never attaches them (RFA:353-357); like the GPU/double variants we DO apply
// them when given.  All float arithmetic is kept in float, in the reference's
// operation order; build with -ffp-contract=off.
#include "xo.h"
#include "xo_internal.h"
#include <algorithm>
#include <cmath>
#include <complex>
#include <cstring>
#include <limits>
#include <vector>

namespace {
const double PI = 3.14159265358979323846;
const int BLOB_TABLE_SIZE_SQRT = 10000;
const double ACCURACY = 0.001;

double bessi0(double x)
{
    double y, ax, ans;
    if ((ax = std::fabs(x)) < 3.75) {
        y = x / 3.75;
        y *= y;
        ans = 1.0 + y * (3.5156229 + y * (3.0899424 + y * (1.2067492 + y * (0.2659732 + y * (0.360768e-1 + y * 0.45813e-2)))));
    } else {
        y = 3.75 / ax;
        ans = (std::exp(ax) / std::sqrt(ax)) * (0.39894228 + y * (0.1328592e-1 + y * (0.225319e-2 + y * (-0.157565e-2 + y * (0.916281e-2 + y * (-0.2057706e-1 + y * (0.2635537e-1 + y * (-0.1647633e-1 + y * 0.392377e-2))))))));
    }
    return ans;
}
double bessi1(double x)
{
    double ax, ans, y;
    if ((ax = std::fabs(x)) < 3.75) {
        y = x / 3.75;
        y *= y;
        ans = ax * (0.5 + y * (0.87890594 + y * (0.51498869 + y * (0.15084934 + y * (0.2658733e-1 + y * (0.301532e-2 + y * 0.32411e-3))))));
    } else {
        y = 3.75 / ax;
        ans = 0.2282967e-1 + y * (-0.2895312e-1 + y * (0.1787654e-1 - y * 0.420059e-2));
        ans = 0.39894228 + y * (-0.3988024e-1 + y * (-0.362018e-2 + y * (0.163801e-2 + y * (-0.1031555e-1 + y * ans))));
        ans *= (std::exp(ax) / std::sqrt(ax));
    }
    return x < 0.0 ? -ans : ans;
}
double bessi2(double x) { return (x == 0) ? 0 : bessi0(x) - ((2 * 1) / x) * bessi1(x); }
double bessi3(double x) { return (x == 0) ? 0 : bessi1(x) - ((2 * 2) / x) * bessi2(x); }
double bessi4(double x) { return (x == 0) ? 0 : bessi2(x) - ((2 * 3) / x) * bessi3(x); }
// half-integer orders (closed forms)
double bessi1_5(double x) { return (x == 0) ? 0 : std::sqrt(2 / (PI * x)) * (std::cosh(x) - (std::sinh(x) / x)); }
double bessj1_5(double x) { return (x == 0) ? 0 : std::sqrt(2 / (PI * x)) * ((std::sin(x) / x) - std::cos(x)); }
double bessi0_5(double x) { return (x == 0) ? 0 : std::sqrt(2 / (PI * x)) * std::sinh(x); }
double bessi2_5(double x) { return (x == 0) ? 0 : bessi0_5(x) - (3 / x) * bessi1_5(x); }
double bessi3_5(double x) { return (x == 0) ? 0 : bessi1_5(x) - (5 / x) * bessi2_5(x); }
double bessj3_5(double x)
{
    if (x == 0) return 0;
    const double rx = std::sqrt(2 / (PI * x));
    return rx * ((15 / (x * x * x) - 6 / x) * std::sin(x) - (15 / (x * x) - 1) * std::cos(x));
}
// Numerical Recipes bessj0 (xmippCore numerical_recipes)
double bessj0(double x)
{
    double ax, z, xx, y, ans, ans1, ans2;
    if ((ax = std::fabs(x)) < 8.0) {
        y = x * x;
        ans1 = 57568490574.0 + y * (-13362590354.0 + y * (651619640.7 + y * (-11214424.18 + y * (77392.33017 + y * (-184.9052456)))));
        ans2 = 57568490411.0 + y * (1029532985.0 + y * (9494680.718 + y * (59272.64853 + y * (267.8532712 + y * 1.0))));
        ans = ans1 / ans2;
    } else {
        z = 8.0 / ax;
        y = z * z;
        xx = ax - 0.785398164;
        ans1 = 1.0 + y * (-0.1098628627e-2 + y * (0.2734510407e-4 + y * (-0.2073370639e-5 + y * 0.2093887211e-6)));
        ans2 = -0.1562499995e-1 + y * (0.1430488765e-3 + y * (-0.6911147651e-5 + y * (0.7621095161e-6 - y * 0.934935152e-7)));
        ans = std::sqrt(0.636619772 / ax) * (std::cos(xx) * ans1 - z * std::sin(xx) * ans2);
    }
    return ans;
}

double kaiser_value(double r, double a, double alpha, int m)
{
    // BLB:37-84
    double rda, rdas, arg, w;
    rda = r / a;
    if (rda <= 1.0) {
        rdas = rda * rda;
        arg = alpha * std::sqrt(1.0 - rdas);
        if (m == 0) w = bessi0(arg) / bessi0(alpha);
        else if (m == 1) { w = std::sqrt(1.0 - rdas); if (alpha != 0.0) w *= bessi1(arg) / bessi1(alpha); }
        else if (m == 2) { w = std::sqrt(1.0 - rdas); w = w * w; if (alpha != 0.0) w *= bessi2(arg) / bessi2(alpha); }
        else if (m == 3) { w = std::sqrt(1.0 - rdas); w = w * w * w; if (alpha != 0.0) w *= bessi3(arg) / bessi3(alpha); }
        else if (m == 4) { w = std::sqrt(1.0 - rdas); w = w * w * w * w; if (alpha != 0.0) w *= bessi4(arg) / bessi4(alpha); }
        else w = std::numeric_limits<double>::quiet_NaN();
    } else w = 0.0;
    return w;
}

double kaiser_Fourier_value(double w, double a, double alpha, int m)
{
    // BLB:144-172 (orders 0 and 2 only)
    double sigma = std::sqrt(std::fabs(alpha * alpha - (2. * PI * a * w) * (2. * PI * a * w)));
    if (m == 2) {
        if (2. * PI * a * w > alpha)
            return std::pow(2. * PI, 3. / 2.) * std::pow(a, 3.) * std::pow(alpha, 2.) * bessj3_5(sigma) / (bessi0(alpha) * std::pow(sigma, 3.5));
        else
            return std::pow(2. * PI, 3. / 2.) * std::pow(a, 3.) * std::pow(alpha, 2.) * bessi3_5(sigma) / (bessi0(alpha) * std::pow(sigma, 3.5));
    } else if (m == 0) {
        if (2 * PI * a * w > alpha)
            return std::pow(2. * PI, 3. / 2.) * std::pow(a, 3) * bessj1_5(sigma) / (bessi0(alpha) * std::pow(sigma, 1.5));
        else
            return std::pow(2. * PI, 3. / 2.) * std::pow(a, 3) * bessi1_5(sigma) / (bessi0(alpha) * std::pow(sigma, 1.5));
    }
    return std::numeric_limits<double>::quiet_NaN();
}

struct Point3D { float x, y, z; };

inline void multiply(const float t[3][3], Point3D &p)
{
    float tmp0 = t[0][0] * p.x + t[0][1] * p.y + t[0][2] * p.z;
    float tmp1 = t[1][0] * p.x + t[1][1] * p.y + t[1][2] * p.z;
    float tmp2 = t[2][0] * p.x + t[2][1] * p.y + t[2][2] * p.z;
    p.x = tmp0; p.y = tmp1; p.z = tmp2;
}
template <typename T> inline bool inRange(T x, T mn, T mx) { return (x > mn) && (x < mx); }
template <typename T, typename U> inline U clampv(U val, T mn, T mx)
{
    U res = val;
    res = (res > mx) ? mx : res;
    res = (res < mn) ? mn : res;
    return res;
}
inline bool getX(float &x, float y, float z, const Point3D &a, const Point3D &b, const Point3D &p0)
{
    // RFA:479-490
    float x0 = p0.x, y0 = p0.y, z0 = p0.z;
    float u = ((z - z0) * a.y + (y0 - y) * a.z) / (a.y * b.z - b.y * a.z);
    float t = (-y0 + y - u * b.y) / (a.y);
    x = x0 + t * a.x + u * b.x;
    return inRange(t, 0.f, 1.f) && inRange(u, 0.f, 1.f);
}

void inv3x3d(const double *A, double *B)
{
    const double a = A[0], b = A[1], c = A[2], d = A[3], e = A[4], f = A[5], g = A[6], h = A[7], i = A[8];
    const double det = a * (e * i - f * h) - b * (d * i - f * g) + c * (d * h - e * g);
    const double id = 1.0 / det;
    B[0] = (e * i - f * h) * id; B[1] = (c * h - b * i) * id; B[2] = (b * f - c * e) * id;
    B[3] = (f * g - d * i) * id; B[4] = (a * i - c * g) * id; B[5] = (c * d - a * f) * id;
    B[6] = (d * h - e * g) * id; B[7] = (b * g - a * h) * id; B[8] = (a * e - b * d) * id;
}
}  // namespace

struct xo_rf {
    xo_rf_params p;
    float blobTableSqrt[BLOB_TABLE_SIZE_SQRT];
    std::vector<double> Fourier_blob_table;
    float blob_radius_f;  // blob.radius as used in float expressions
    int mvX, mvYZ;        // current maxVolumeIndexX / YZ (X halves after mirror)
    bool cropped;
    std::vector<std::complex<float>> tempVolume;  // [z][y][x], (mvYZ+1)^2 * (mvX+1)
    std::vector<float> tempWeights;
    inline size_t idx(int x, int y, int z) const
    {
        return ((size_t)z * (mvYZ + 1) + y) * (mvX + 1) + x;
    }
};

namespace {
// RFA:627-700
inline void processVoxelBlob(xo_rf *R, int x, int y, int z, const float transform[3][3],
                             float maxDistanceSqr, const std::complex<float> *img,
                             const float *CTF, const float *modulator, float dataWeight, int imgXS,
                             int imgYS)
{
    Point3D imgPos;
    const int maxVolumeIndexX = R->mvX, maxVolumeIndexYZ = R->mvYZ;
    imgPos.x = x - maxVolumeIndexX / 2;
    imgPos.y = y - maxVolumeIndexYZ / 2;
    imgPos.z = z - maxVolumeIndexYZ / 2;
    if ((imgPos.x * imgPos.x + imgPos.y * imgPos.y + imgPos.z * imgPos.z) > maxDistanceSqr) return;
    multiply(transform, imgPos);
    imgPos.y += maxVolumeIndexYZ / 2;
    // blob.radius is a double member in the reference: the products below are
    // evaluated in double and then narrowed to float exactly as written there.
    const double blobRadius = R->p.blob_radius;
    float radiusSqr = blobRadius * blobRadius;
    float zSqr = imgPos.z * imgPos.z;
    if (zSqr > radiusSqr) return;
    int minX = std::ceil(imgPos.x - blobRadius);
    int maxX = std::floor(imgPos.x + blobRadius);
    int minY = std::ceil(imgPos.y - blobRadius);
    int maxY = std::floor(imgPos.y + blobRadius);
    minX = std::max(minX, 0);
    minY = std::max(minY, 0);
    maxX = std::min(maxX, imgXS - 1);
    maxY = std::min(maxY, imgYS - 1);
    std::complex<float> *targetVolume = &R->tempVolume[R->idx(x, y, z)];
    float *targetWeight = &R->tempWeights[R->idx(x, y, z)];
    const float iDeltaSqrt = R->p.iDeltaSqrt;
    if (CTF) {
        for (int i = minY; i <= maxY; i++) {
            float ySqr = (imgPos.y - i) * (imgPos.y - i);
            float yzSqr = ySqr + zSqr;
            if (yzSqr > radiusSqr) continue;
            for (int j = minX; j <= maxX; j++) {
                float xD = imgPos.x - j;
                float distanceSqr = xD * xD + yzSqr;
                if (distanceSqr > radiusSqr) continue;
                float wCTF = CTF[(size_t)i * imgXS + j];
                float wModulator = modulator[(size_t)i * imgXS + j];
                int aux = (int)(distanceSqr * iDeltaSqrt + 0.5f);
                float wBlob = R->blobTableSqrt[aux];
                float weight = wBlob * wModulator * dataWeight;
                *targetWeight += weight;
                *targetVolume += img[(size_t)i * imgXS + j] * weight * wCTF;
            }
        }
    } else {
        for (int i = minY; i <= maxY; i++) {
            float ySqr = (imgPos.y - i) * (imgPos.y - i);
            float yzSqr = ySqr + zSqr;
            if (yzSqr > radiusSqr) continue;
            for (int j = minX; j <= maxX; j++) {
                float xD = imgPos.x - j;
                float distanceSqr = xD * xD + yzSqr;
                if (distanceSqr > radiusSqr) continue;
                int aux = (int)(distanceSqr * iDeltaSqrt + 0.5f);
                float wBlob = R->blobTableSqrt[aux];
                float weight = wBlob * dataWeight;
                *targetWeight += weight;
                *targetVolume += img[(size_t)i * imgXS + j] * weight;
            }
        }
    }
}

// RFA:595-625
inline void processVoxel(xo_rf *R, int x, int y, int z, const float transform[3][3],
                         float maxDistanceSqr, const std::complex<float> *img, const float *CTF,
                         const float *modulator, float dataWeight, int imgXS, int imgYS)
{
    Point3D imgPos;
    float wBlob = 1.f, wCTF = 1.f, wModulator = 1.f;
    const int maxVolumeIndexX = R->mvX, maxVolumeIndexYZ = R->mvYZ;
    imgPos.x = x - maxVolumeIndexX / 2;
    imgPos.y = y - maxVolumeIndexYZ / 2;
    imgPos.z = z - maxVolumeIndexYZ / 2;
    if (imgPos.x * imgPos.x + imgPos.y * imgPos.y + imgPos.z * imgPos.z > maxDistanceSqr) return;
    multiply(transform, imgPos);
    int imgX = clampv((int)(imgPos.x + 0.5f), 0, imgXS - 1);
    int imgY = clampv((int)(imgPos.y + 0.5f + maxVolumeIndexYZ / 2), 0, imgYS - 1);
    if (CTF) {
        wCTF = CTF[(size_t)imgY * imgXS + imgX];
        wModulator = modulator[(size_t)imgY * imgXS + imgX];
    }
    float weight = wBlob * wModulator * dataWeight;
    R->tempVolume[R->idx(x, y, z)] += img[(size_t)imgY * imgXS + imgX] * weight * wCTF;
    R->tempWeights[R->idx(x, y, z)] += weight;
}

void processProjection(xo_rf *R, const std::complex<float> *img, const float *CTF,
                       const float *modulator, float dataWeight, const float transform[3][3],
                       const float transformInv[3][3])
{
    // RFA:710-763
    const int maxVolumeIndexX = R->mvX, maxVolumeIndexYZ = R->mvYZ;
    const bool useFast = R->p.useFast != 0;
    int imgSizeX = maxVolumeIndexX / 2;  // cropAndShift: sizeX = maxVolumeIndexX/2 (RFA:273)
    int imgSizeY = maxVolumeIndexYZ;
    const double blobRadius = R->p.blob_radius;
    const float maxDistanceSqr = (imgSizeX + (useFast ? 0.f : blobRadius)) * (imgSizeX + (useFast ? 0.f : blobRadius));
    Point3D origin = {maxVolumeIndexX / 2.f, maxVolumeIndexYZ / 2.f, maxVolumeIndexYZ / 2.f};
    Point3D u, v;
    Point3D AABB[2];
    Point3D cuboid[8];
    // createProjectionCuboid RFA:430-442
    {
        float sizeX = imgSizeX, sizeY = imgSizeY, blobSize = useFast ? 0.f : blobRadius;
        float halfY = sizeY / 2.0f;
        cuboid[0].x = cuboid[3].x = cuboid[4].x = cuboid[7].x = 0.f - blobSize;
        cuboid[1].x = cuboid[2].x = cuboid[5].x = cuboid[6].x = sizeX + blobSize;
        cuboid[0].y = cuboid[1].y = cuboid[4].y = cuboid[5].y = -(halfY + blobSize);
        cuboid[2].y = cuboid[3].y = cuboid[6].y = cuboid[7].y = halfY + blobSize;
        cuboid[0].z = cuboid[1].z = cuboid[2].z = cuboid[3].z = 0.f + blobSize;
        cuboid[4].z = cuboid[5].z = cuboid[6].z = cuboid[7].z = 0.f - blobSize;
    }
    for (int i = 0; i < 8; i++) multiply(transform, cuboid[i]);
    for (int i = 0; i < 8; i++) { cuboid[i].x += origin.x; cuboid[i].y += origin.y; cuboid[i].z += origin.z; }
    // computeAABB RFA:499-522 (max seeded with numeric_limits<float>::min(), kept)
    {
        AABB[0].x = AABB[0].y = AABB[0].z = std::numeric_limits<float>::max();
        AABB[1].x = AABB[1].y = AABB[1].z = std::numeric_limits<float>::min();
        for (int i = 0; i < 8; i++) {
            Point3D tmp = cuboid[i];
            if (AABB[0].x > tmp.x) AABB[0].x = tmp.x;
            if (AABB[0].y > tmp.y) AABB[0].y = tmp.y;
            if (AABB[0].z > tmp.z) AABB[0].z = tmp.z;
            if (AABB[1].x < tmp.x) AABB[1].x = tmp.x;
            if (AABB[1].y < tmp.y) AABB[1].y = tmp.y;
            if (AABB[1].z < tmp.z) AABB[1].z = tmp.z;
        }
        float minX = 0, minY = 0, minZ = 0, maxX = maxVolumeIndexX, maxY = maxVolumeIndexYZ, maxZ = maxVolumeIndexYZ;
        if (AABB[0].x < minX) AABB[0].x = minX;
        if (AABB[0].y < minY) AABB[0].y = minY;
        if (AABB[0].z < minZ) AABB[0].z = minZ;
        if (AABB[1].x > maxX) AABB[1].x = maxX;
        if (AABB[1].y > maxY) AABB[1].y = maxY;
        if (AABB[1].z > maxZ) AABB[1].z = maxZ;
    }
    // getVectors RFA:258-269
    {
        float x0 = cuboid[0].x, y0 = cuboid[0].y, z0 = cuboid[0].z;
        u.x = cuboid[1].x - x0; u.y = cuboid[1].y - y0; u.z = cuboid[1].z - z0;
        v.x = cuboid[3].x - x0; v.y = cuboid[3].y - y0; v.z = cuboid[3].z - z0;
    }
    int minY, minZ, maxY, maxZ;
    minZ = std::floor(AABB[0].z);
    minY = std::floor(AABB[0].y);
    maxZ = std::ceil(AABB[1].z);
    maxY = std::ceil(AABB[1].y);
    for (int z = minZ; z <= maxZ; z++) {
        for (int y = minY; y <= maxY; y++) {
            if (useFast) {
                float hitX;
                if (getX(hitX, y, z, u, v, *cuboid)) {
                    int x = (int)(hitX + 0.5f);
                    processVoxel(R, x, y, z, transformInv, maxDistanceSqr, img, CTF, modulator, dataWeight, imgSizeX, imgSizeY);
                }
            } else {
                float x1, x2;
                bool hit1 = getX(x1, y, z, u, v, *cuboid);
                bool hit2 = getX(x2, y, z, u, v, *(cuboid + 4));
                if (hit1 || hit2) {
                    x1 = clampv(x1, 0, maxVolumeIndexX);
                    x2 = clampv(x2, 0, maxVolumeIndexX);
                    float lower = std::min(x1, x2);
                    float upper = std::max(x1, x2);
                    for (int x = std::floor(lower); x <= std::ceil(upper); x++)
                        processVoxelBlob(R, x, y, z, transformInv, maxDistanceSqr, img, CTF, modulator, dataWeight, imgSizeX, imgSizeY);
                }
            }
        }
    }
}

template <typename T> T identityf(T v) { return v; }
std::complex<float> conjf_(std::complex<float> v) { return std::conj(v); }

template <typename T, typename F>
void mirrorAndCrop(xo_rf *R, std::vector<T> &input, F f, int newX)
{
    // RFA:861-887 ; called with maxVolumeIndexX already halved
    const int mvYZ = R->mvYZ;
    std::vector<T> output((size_t)(mvYZ + 1) * (mvYZ + 1) * (newX + 1), T(0));
    auto oidx = [&](int x, int y, int z) { return ((size_t)z * (mvYZ + 1) + y) * (newX + 1) + x; };
    auto iidx = [&](int x, int y, int z) { return ((size_t)z * (mvYZ + 1) + y) * (mvYZ + 1) + x; };
    for (int z = 0; z <= mvYZ; z++)
        for (int y = 0; y <= mvYZ; y++)
            for (int x = 0; x <= mvYZ; x++) {
                if (x < newX) {
                    int n0 = mvYZ - x, n1 = mvYZ - y, n2 = mvYZ - z;
                    output[oidx(n0 - newX, n1, n2)] += f(input[iidx(x, y, z)]);
                } else {
                    output[oidx(x - newX, y, z)] += input[iidx(x, y, z)];
                }
            }
    input.swap(output);
}

template <typename T>
void applyBlob(xo_rf *R, std::vector<T> &input, float blobSize)
{
    // RFA:793-831
    const int mvX = R->mvX, mvYZ = R->mvYZ;
    float blobSizeSqr = blobSize * blobSize;
    int blob = std::floor(blobSize);
    std::vector<T> output(input.size(), T(0));
    const float iDeltaSqrt = R->p.iDeltaSqrt;
#pragma omp parallel for
    for (int i = 0; i <= mvYZ; i++)
        for (int j = 0; j <= mvYZ; j++)
            for (int k = 0; k <= mvX; k++) {
                T tmp = (T)0;
                for (int z = std::max(0, i - blob); z <= std::min(mvYZ, i + blob); z++) {
                    float dZSqr = (i - z) * (i - z);
                    for (int y = std::max(0, j - blob); y <= std::min(mvYZ, j + blob); y++) {
                        float dYSqr = (j - y) * (j - y);
                        for (int x = std::max(0, k - blob); x <= std::min(mvX, k + blob); x++) {
                            float dXSqr = (k - x) * (k - x);
                            float distanceSqr = dZSqr + dYSqr + dXSqr;
                            if (distanceSqr > blobSizeSqr) continue;
                            int aux = (int)(distanceSqr * iDeltaSqrt + 0.5f);
                            float tmpWeight = R->blobTableSqrt[aux];
                            tmp += tmpWeight * input[R->idx(x, y, z)];
                        }
                    }
                }
                output[R->idx(k, j, i)] = tmp;
            }
    input.swap(output);
}
}  // namespace

extern "C" {

double xo_kaiser_value(double r, double a, double alpha, int m) { return kaiser_value(r, a, alpha, m); }
double xo_kaiser_fourier_value(double w, double a, double alpha, int m) { return kaiser_Fourier_value(w, a, alpha, m); }
double xo_bessi0(double x) { return bessi0(x); }
double xo_bessi1(double x) { return bessi1(x); }

void xo_euler_matrix(double alpha, double beta, double gamma, double *A)
{
    // xmippCore Euler_angles2matrix; closed form test_geometry_main.cpp:46-65,
    // in-tree copy reconstruction_cuda/cuda_fourier_projection.cpp:38-73
    double ca, sa, cb, sb, cg, sg, cc, cs, sc, ss;
    alpha = alpha * PI / 180.; beta = beta * PI / 180.; gamma = gamma * PI / 180.;
    ca = std::cos(alpha); cb = std::cos(beta); cg = std::cos(gamma);
    sa = std::sin(alpha); sb = std::sin(beta); sg = std::sin(gamma);
    cc = cb * ca; cs = cb * sa; sc = sb * ca; ss = sb * sa;
    A[0] = cg * cc - sg * sa; A[1] = cg * cs + sg * ca; A[2] = -cg * sb;
    A[3] = -sg * cc - cg * sa; A[4] = -sg * cs + cg * ca; A[5] = sg * sb;
    A[6] = sc; A[7] = ss; A[8] = cb;
}

/* ---- CTF ---- */
void xo_ctf_defaults(xo_ctf_params *p)
{
    // ctf.cpp:365-388 readFromMdRow defaults / clear()
    std::memset(p, 0, sizeof(*p));
    p->Tm = 1; p->kV = 100; p->K = 1;
}
double xo_ctf_lambda(const xo_ctf_params *p)
{
    double local_kV = p->kV * 1e3;
    return 12.2643247 / std::sqrt(local_kV * (1. + 0.978466e-6 * local_kV));
}
double xo_ctf_value_pure_nok(const xo_ctf_params *p, double X, double Y)
{
    // produceSideInfo ctf.cpp:645-679,1392-1402
    const double local_Cs = p->Cs * 1e7, local_Ca = p->Ca * 1e7, local_ispr = p->ispr * 1e6;
    const double lambda = xo_ctf_lambda(p);
    const double K1 = PI * lambda;
    const double K2 = PI / 2 * local_Cs * lambda * lambda * lambda;
    const double K3 = std::pow(0.25 * PI * local_Ca * lambda * (p->espr / p->kV + 2 * local_ispr), 2) / std::log(2.0);
    const double K5 = PI * p->DeltaF * lambda;
    const double K6 = PI * PI * p->alpha * p->alpha;
    const double K7 = local_Cs * lambda * lambda;
    const double Ksin = std::sqrt(1 - p->Q0 * p->Q0);
    const double Kcos = p->Q0;
    const double rad_azimuth = p->azimuthal_angle * PI / 180.;
    const double defocus_average = -(p->DeltafU + p->DeltafV) * 0.5;
    const double defocus_deviation = -(p->DeltafU - p->DeltafV) * 0.5;
    // precomputeValues(X,Y) ctf.h:1002-1029
    const double ang = std::atan2(Y, X);
    const double u2 = X * X + Y * Y;
    const double u = std::sqrt(u2);
    const double u4 = u2 * u2;
    double deltaf;
    if (std::fabs(X) < XO_EQUAL_ACCURACY && std::fabs(Y) < XO_EQUAL_ACCURACY) deltaf = 0;
    else {
        double ellipsoid_ang = ang - rad_azimuth;
        double cos_ellipsoid_ang_2 = std::cos(2 * ellipsoid_ang);
        deltaf = (defocus_average + defocus_deviation * cos_ellipsoid_ang_2);
    }
    // getValuePureAt ctf.h:452-496
    double VPP = 0.0;
    double check_VPP = std::round(p->VPP_radius * 1000);
    if (check_VPP != 0) VPP = -p->phase_shift * (1 - std::exp(-u2 / (2 * std::pow(p->VPP_radius, 2.0))));
    double argument = VPP + K1 * deltaf * u2 + K2 * u4;
    double sine_part = std::sin(argument);
    double cosine_part = std::cos(argument);
    double Eespr = std::exp(-K3 * u4);
    double EdeltaF = bessj0(K5 * u2);
    double xs = u * p->DeltaR;
    double EdeltaR = (xs == 0) ? 1.0 : std::sin(PI * xs) / (PI * xs);  // SINC
    double aux = (K7 * u2 * u + deltaf * u);
    double Ealpha = std::exp(-K6 * aux * aux);
    double E = Eespr * EdeltaF * EdeltaR * Ealpha + p->envR0 + p->envR1 * u + p->envR2 * u2;
    if (E < 0) E = 0;
    double pure = -p->K * (Ksin * sine_part - Kcos * cosine_part) * E;
    return p->K * pure;  // getValuePureNoKAt ctf.h:499-502 (multiplies by K; quirk kept)
}

/* ---- reconstruction ---- */
xo_rf *xo_rf_create(xo_rf_params *p)
{
    xo_rf *R = new xo_rf;
    // RFA:175-257
    const int Xdim = p->imgSize;
    p->paddedImgSize = Xdim * p->padding_vol;
    size_t conserveRows = (size_t)std::ceil((double)p->paddedImgSize * p->maxResolution * 2.0);
    conserveRows = (size_t)std::ceil((double)conserveRows / 2.0);
    p->maxVolumeIndexX = p->maxVolumeIndexYZ = 2 * conserveRows;
    R->Fourier_blob_table.resize(BLOB_TABLE_SIZE_SQRT);
    double blobFourier_radius = p->blob_radius / (p->padding_vol * Xdim);
    double blobnormalized_radius = p->blob_radius / ((double)p->padding_proj / p->padding_vol);
    double deltaSqrt = (p->blob_radius * p->blob_radius) / (BLOB_TABLE_SIZE_SQRT - 1);
    double deltaFourier = (std::sqrt(3.) * Xdim / 2.) / (BLOB_TABLE_SIZE_SQRT - 1);
    double iw0 = 1.0 / kaiser_Fourier_value(0.0, blobnormalized_radius, p->blob_alpha, p->blob_order);
    double padXdim3 = p->padding_vol * Xdim;
    padXdim3 = padXdim3 * padXdim3 * padXdim3;
    double blobTableSize = p->blob_radius * std::sqrt(1. / (BLOB_TABLE_SIZE_SQRT - 1));
    for (int i = 0; i < BLOB_TABLE_SIZE_SQRT; i++) {
        R->blobTableSqrt[i] = kaiser_value(blobTableSize * std::sqrt((double)i), p->blob_radius, p->blob_alpha, p->blob_order) * iw0;
        R->Fourier_blob_table[i] = kaiser_Fourier_value(deltaFourier * i, blobFourier_radius, p->blob_alpha, p->blob_order) * padXdim3 * iw0;
    }
    p->iDeltaSqrt = 1 / deltaSqrt;
    p->iDeltaFourier = 1 / deltaFourier;
    R->p = *p;
    R->mvX = p->maxVolumeIndexX;
    R->mvYZ = p->maxVolumeIndexYZ;
    R->cropped = false;
    xo_rf_reset(R);
    return R;
}
void xo_rf_destroy(xo_rf *R) { delete R; }
const float *xo_rf_blob_table_sqrt(const xo_rf *R) { return R->blobTableSqrt; }
const double *xo_rf_fourier_blob_table(const xo_rf *R) { return R->Fourier_blob_table.data(); }
float *xo_rf_temp_volume(xo_rf *R) { return reinterpret_cast<float *>(R->tempVolume.data()); }
float *xo_rf_temp_weights(xo_rf *R) { return R->tempWeights.data(); }
void xo_rf_reset(xo_rf *R)
{
    R->mvX = R->p.maxVolumeIndexX;
    R->mvYZ = R->p.maxVolumeIndexYZ;
    R->cropped = false;
    const size_t n = (size_t)(R->mvYZ + 1) * (R->mvYZ + 1) * (R->mvX + 1);
    R->tempVolume.assign(n, std::complex<float>(0, 0));
    R->tempWeights.assign(n, 0.f);
}

void xo_rf_prepare_image(const xo_rf *R, const double *img, float *fft_out)
{
    // RFA:332-351 + cropAndShift RFA:271-298
    const int D = R->p.imgSize, P = R->p.paddedImgSize;
    std::vector<double> padded((size_t)P * P, 0.0);
    // A2D_ELEM(localPaddedImg,i,j) = A2D_ELEM(mProj,i,j) with Xmipp origins, then CenterFFT(,true)
    const int s0 = xo::first_xmipp_index(D), p0 = xo::first_xmipp_index(P);
    const int sh = P / 2;
    for (int i = 0; i < D; ++i)
        for (int j = 0; j < D; ++j) {
            int pi = (i + s0 - p0), pj = (j + s0 - p0);  // physical index in padded
            padded[(size_t)((pi + sh) % P) * P + (pj + sh) % P] = img[(size_t)i * D + j];
        }
    const int xh = P / 2 + 1;
    std::vector<double> F((size_t)P * xh * 2);
    xo_fft2d_r2c(padded.data(), P, P, F.data());
    const int sizeX = R->p.maxVolumeIndexX / 2, sizeY = R->p.maxVolumeIndexYZ;
    std::complex<float> *res = reinterpret_cast<std::complex<float> *>(fft_out);
    for (size_t n = 0; n < (size_t)sizeX * sizeY; ++n) res[n] = std::complex<float>(0, 0);
    const int halfY = P / 2;
    const double maxResolutionSqr = R->p.maxResolution * R->p.maxResolution;
    for (int i = 0; i < P; i++)
        for (int j = 0; j < sizeX; j++)
            if (i < sizeX || i >= (P - sizeX)) {
                double t0 = xo_fft_idx2digfreq(j, P), t1 = xo_fft_idx2digfreq(i, P);
                if (t0 * t0 + t1 * t1 > maxResolutionSqr) continue;
                int myPadI = (i < halfY) ? i + sizeX : i - P + sizeX;
                res[(size_t)myPadI * sizeX + j] = std::complex<float>(F[2 * ((size_t)i * xh + j)], F[2 * ((size_t)i * xh + j) + 1]);
            }
}

void xo_rf_ctf_arrays(const xo_rf *R, const xo_ctf_params *ctf, float *CTF, float *modulator)
{
    // RFA:548-592
    const int XS = R->p.maxVolumeIndexX / 2, YS = R->p.maxVolumeIndexYZ;
    const int P = R->p.paddedImgSize;
    for (int y = 0; y < YS; y++) {
        float freqY = (y - (P / 2.f)) / (float)P;
        for (int x = 0; x < XS; x++) {
            float modulatorVal = 1.f;
            float freqX = xo_fft_idx2digfreq(x, P);
            float CTFVal = xo_ctf_value_pure_nok(ctf, freqX * R->p.iTs, freqY * R->p.iTs);
            if (std::isnan(CTFVal)) {
                if ((x == 0) && (y == 0)) modulatorVal = CTFVal = 1.0;
                else modulatorVal = CTFVal = 0.0;
            }
            if (std::fabs(CTFVal) < R->p.minCTF) {
                modulatorVal = std::fabs(CTFVal);
                CTFVal = (CTFVal >= 0) ? 1 : -1;  // SGN (xmippCore: ((x) >= 0) ? 1 : -1)
            } else CTFVal = 1.0 / CTFVal;
            if (R->p.isPhaseFlipped) CTFVal = std::fabs(CTFVal);
            CTF[(size_t)y * XS + x] = CTFVal;
            modulator[(size_t)y * XS + x] = modulatorVal;
        }
    }
}

void xo_rf_insert(xo_rf *R, const float *fft, const float *ctf, const float *modulator,
                  const double *localAInv, const double *Rsym, float weight)
{
    // processBuffer RFA:953-965
    double A_SL[9], A_SLInv[9];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            double s = 0;
            for (int k = 0; k < 3; ++k) s += Rsym[i * 3 + k] * localAInv[k * 3 + j];
            A_SL[i * 3 + j] = s;
        }
    inv3x3d(A_SL, A_SLInv);
    float transf[3][3], transfInv[3][3];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) { transf[i][j] = A_SL[i * 3 + j]; transfInv[i][j] = A_SLInv[i * 3 + j]; }
    processProjection(R, reinterpret_cast<const std::complex<float> *>(fft), ctf, modulator, weight, transf, transfInv);
}

void xo_rf_mirror_and_crop(xo_rf *R)
{
    const int newX = R->mvYZ / 2;
    mirrorAndCrop(R, R->tempWeights, identityf<float>, newX);
    mirrorAndCrop(R, R->tempVolume, conjf_, newX);
    R->mvX = newX;
    R->cropped = true;
}

void xo_rf_hermitian_and_weights(xo_rf *R)
{
    // forceHermitianSymmetry RFA:889-906
    const int mvYZ = R->mvYZ, mvX = R->mvX;
    {
        int x = 0;
        for (int z = 0; z <= mvYZ; z++)
            for (int y = 0; y <= mvYZ / 2; y++) {
                int n0 = x, n1 = mvYZ - y, n2 = mvYZ - z;
                std::complex<float> tmp1 = 0.5f * (R->tempVolume[R->idx(n0, n1, n2)] + std::conj(R->tempVolume[R->idx(x, y, z)]));
                float tmp2 = 0.5f * (R->tempWeights[R->idx(n0, n1, n2)] + R->tempWeights[R->idx(x, y, z)]);
                R->tempVolume[R->idx(n0, n1, n2)] = tmp1;
                R->tempVolume[R->idx(x, y, z)] = std::conj(tmp1);
                R->tempWeights[R->idx(n0, n1, n2)] = R->tempWeights[R->idx(x, y, z)] = tmp2;
            }
    }
    // processWeights RFA:908-924
    float corr2D_3D = std::pow(R->p.padding_proj, 2.) / (R->p.imgSize * std::pow(R->p.padding_vol, 3.));
    for (int z = 0; z <= mvYZ; z++)
        for (int y = 0; y <= mvYZ; y++)
            for (int x = 0; x <= mvX; x++) {
                float weight = R->tempWeights[R->idx(x, y, z)];
                if (weight > ACCURACY) R->tempVolume[R->idx(x, y, z)] *= corr2D_3D / weight;
                else R->tempVolume[R->idx(x, y, z)] = 0;
            }
}

void xo_rf_finish(xo_rf *R, double *vol_out)
{
    // finishComputations RFA:1002-1055 (expects mirror_and_crop done)
    if (R->p.useFast) {
        applyBlob(R, R->tempVolume, (float)R->p.blob_radius);
        applyBlob(R, R->tempWeights, (float)R->p.blob_radius);
    }
    xo_rf_hermitian_and_weights(R);
    const int P = R->p.paddedImgSize, D = R->p.imgSize;
    const int xh = P / 2 + 1;
    std::vector<std::complex<double>> VoutFourier((size_t)P * P * xh, std::complex<double>(0, 0));
    // convertToExpectedSpace RFA:834-851
    {
        const int size = R->mvYZ, halfSize = size / 2;
        for (int z = 0; z <= size; z++)
            for (int y = 0; y <= size; y++)
                for (int x = 0; x <= halfSize; x++) {
                    int n0 = x;
                    int n1 = (y < halfSize) ? P - halfSize + y : y - halfSize;
                    int n2 = (z < halfSize) ? P - halfSize + z : z - halfSize;
                    std::complex<float> v = R->tempVolume[R->idx(x, y, z)];
                    VoutFourier[((size_t)n2 * P + n1) * xh + n0] += std::complex<double>(v.real(), v.imag());
                }
    }
    std::vector<double> Vout((size_t)P * P * P);
    xo_fft3d_c2r(reinterpret_cast<const double *>(VoutFourier.data()), P, P, P, Vout.data());
    VoutFourier.clear(); VoutFourier.shrink_to_fit();
    // CenterFFT(Vout,false) then window to imgSize with Xmipp origin:
    // logical coordinate l (in [-D/2, D/2)) of the centred array sits at physical l + P/2;
    // CenterFFT backward shifts by -P/2: centred[p] = raw[(p + P/2) mod P]
    const int s0 = xo::first_xmipp_index(D);
    const double pad_rel0 = ((double)R->p.padding_proj / R->p.padding_vol);
    const double pad_relation = pad_rel0 * pad_rel0 * pad_rel0;
    const double ipad_relation = 1.0 / pad_relation;
    double meanFactor2 = 0;
    const int pc = P / 2;  // physical position of logical 0 in the centred padded volume is -FIRST = P/2
    const double iDeltaFourier = R->p.iDeltaFourier;  // float member in the reference
    for (int k = 0; k < D; ++k)
        for (int i = 0; i < D; ++i)
            for (int j = 0; j < D; ++j) {
                const int lk = k + s0, li = i + s0, lj = j + s0;
                const int pk = lk + pc, pi = li + pc, pj = lj + pc;  // in centred array
                const int rk = (pk + P / 2) % P, ri = (pi + P / 2) % P, rj = (pj + P / 2) % P;
                double val = Vout[((size_t)rk * P + ri) * P + rj];
                double radius = std::sqrt((double)(lk * lk + li * li + lj * lj));
                double aux = radius * iDeltaFourier;
                double factor = R->Fourier_blob_table[(int)std::floor(aux + 0.5)];  // ROUND
                double xs = radius / (2 * D);
                double sinc = (xs == 0) ? 1.0 : std::sin(PI * xs) / (PI * xs);
                double factor2 = std::pow(sinc, 2);
                val /= (ipad_relation * factor2 * factor);
                meanFactor2 += factor2;
                vol_out[((size_t)k * D + i) * D + j] = val;
            }
    meanFactor2 /= (double)D * D * D;
    for (size_t n = 0; n < (size_t)D * D * D; ++n) vol_out[n] *= meanFactor2;
}

}  // extern "C"
