// sampling_gen.h -- generation of the projection-direction sampling (host side of
// xmipp_angular_project_library): quasi-uniform points on the sphere from a subdivided icosahedron
// (Baumgardner 1995), reduction to the asymmetric unit, neighbourhoods of the experimental images.
//
// Restates libraries/data/sampling.cpp: constructor vertices (L90-101), setSampling (L121-132),
// computeSamplingPoints (L155-584), fillEdge / fillDistance (L606-676), removeRedundantPoints
// (L702-1216, every group it implements), fillLRRepository (L2216-2234),
// fillExpDataProjectionDirectionByLR (L2256-2290), removePointsFarAwayFromExperimentalData (L1932-1958),
// computeNeighbors (L1715-1860), findClosestSamplingPoint (L1991-2098), createAsymUnitFile (L1440-1488), saveSamplingFile (L1495-1583).
// Pinned by the reference's own fixtures (resources/test/sampling/*.xmd, used by
// function_tests/test_sampling_main.cpp:128-179), copied as data under tests/golden/sampling/.
#ifndef XH_SAMPLING_GEN_H
#define XH_SAMPLING_GEN_H
#include "minicore.h"
#include <array>

namespace mc {
typedef std::array<double, 3> Vec3;

struct SamplingGen {
    static constexpr double cte_w = 1.107149;   // sampling.h:35 (angle between icosahedron vertices)
    double sampling_rate_rad = 0, cos_neighborhood_radius = -1.01;
    size_t number_of_samples = 0, numberSamplesAsymmetricUnit = 0;
    std::vector<Vec3> sampling_points_vector, sampling_points_angles;
    std::vector<Vec3> no_redundant_sampling_points_vector, no_redundant_sampling_points_angles;
    std::vector<size_t> no_redundant_sampling_points_index;
    std::vector<std::vector<double>> R_repository;            // 3x3 row-major, identity first
    std::vector<Vec3> exp_data_projection_direction_by_L_R;
    std::vector<std::string> exp_data_fileNames;
    std::vector<std::vector<size_t>> my_neighbors;

    static const Vec3 &vertex(int i)
    {
        static const Vec3 v[12] = {
            {0., 0., 1.},
            {0.723606900230461, -0.525731185781806, 0.447213343087301},
            {0.723606900230461, 0.525731185781806, 0.447213343087301},
            {-0.276393239417711, 0.850650928976665, 0.447213343087301},
            {-0.8944273172062, 0., 0.447213343087301},
            {-0.276393239417711, -0.850650928976665, 0.447213343087301},
            {0.8944273172062, 0., -0.447213343087301},
            {0.276393242471372, 0.850650927984471, -0.447213343087301},
            {-0.723606898343194, 0.525731188379405, -0.447213343087301},
            {-0.723606898343194, -0.525731188379405, -0.447213343087301},
            {0.276393242471372, -0.850650927984471, -0.447213343087301},
            {0., 0., -1.}};
        return v[i];
    }
    static double dot(const Vec3 &a, const Vec3 &b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
    static Vec3 slerpNormalised(const Vec3 &s, const Vec3 &e, double gamma, double upsilon, double &beta)
    {
        const double alpha = std::sin((1. - gamma) * upsilon) / std::sin(upsilon);
        beta = std::sin(gamma * upsilon) / std::sin(upsilon);
        Vec3 v = {alpha * s[0] + beta * e[0], alpha * s[1] + beta * e[1], alpha * s[2] + beta * e[2]};
        const double n = std::sqrt(dot(v, v));
        v[0] /= n; v[1] /= n; v[2] /= n;
        return v;
    }

    void setSampling(double samplingDeg)
    {
        sampling_rate_rad = samplingDeg * M_PI / 180.;
        number_of_samples = (size_t)std::floor(cte_w / sampling_rate_rad + 0.5) + 1;     // ROUND(.)+1
        if (number_of_samples < 3)
            REPORT_ERROR(ERR_ARG_INCORRECT, "maximum value of angular sampling rate is " + std::to_string(cte_w * 0.5 * 180. / M_PI));
    }
    void setNeighborhoodRadius(double deg)
    {
        if (deg < 0) cos_neighborhood_radius = -1.01;
        else if (deg > 180.001) REPORT_ERROR(ERR_ARG_INCORRECT, "Neighborhood can not be greater than 180");
        else cos_neighborhood_radius = std::cos(deg * M_PI / 180.);
    }

    // points along the arc s -> e, first corner skipped; the last one too when it closes on a corner already taken
    void fillEdge(const Vec3 &s, const Vec3 &e, std::vector<Vec3> &edge, bool endFlag) const
    {
        const double upsilon = std::acos(dot(s, e));
        for (size_t i1 = 1; i1 < number_of_samples; i1++) {
            double beta;
            const Vec3 v = slerpNormalised(s, e, (double)i1 / (double)(number_of_samples - 1), upsilon, beta);
            if (beta > 0.9999 && endFlag) continue;
            edge.push_back(v);
        }
    }
    static bool inRange(const Vec3 &v, bool half, double min_z, double max_z) { return !((half && v[2] < 0.0) || v[2] < min_z || v[2] > max_z); }
    void fillDistance(const Vec3 &s, const Vec3 &e, int n, bool half, double min_z, double max_z)
    {
        const double upsilon = std::acos(dot(s, e));
        for (int i1 = 1; i1 < n; i1++) {
            double beta;
            const Vec3 v = slerpNormalised(s, e, (double)i1 / (double)n, upsilon, beta);
            if (inRange(v, half, min_z, max_z)) sampling_points_vector.push_back(v);
        }
    }

    void computeSamplingPoints(bool only_half_sphere, double max_tilt, double min_tilt)
    {
        sampling_points_angles.clear();
        sampling_points_vector.clear();
        if (min_tilt > max_tilt || min_tilt < 0 || max_tilt < 0 || max_tilt > 180.)
            REPORT_ERROR(ERR_ARG_INCORRECT, "tilt angles cannot be negative or min_tilt > max_tilt");
        double max_z = std::cos(M_PI * max_tilt / 180.), min_z = std::cos(M_PI * min_tilt / 180.);
        if (min_z > max_z) std::swap(min_z, max_z);
        // The 20 faces come as 10 spherical rhombi {apex, far corner, left, right}: the "start" edge runs
        // apex -> left -> (far -> left), the "end" edge apex -> right -> (far -> right)  (L192-337)
        static const int rhombi[10][4] = {{0, 6, 1, 2}, {0, 7, 2, 3}, {0, 8, 3, 4}, {0, 9, 4, 5}, {0, 10, 5, 1},
                                          {11, 5, 10, 9}, {11, 4, 9, 8}, {11, 3, 8, 7}, {11, 2, 7, 6}, {11, 1, 6, 10}};
        std::vector<Vec3> edgeStart, edgeEnd;
        for (const auto &r : rhombi) {
            fillEdge(vertex(r[0]), vertex(r[2]), edgeStart, false);
            fillEdge(vertex(r[1]), vertex(r[2]), edgeStart, true);
            fillEdge(vertex(r[0]), vertex(r[3]), edgeEnd, false);
            fillEdge(vertex(r[1]), vertex(r[3]), edgeEnd, true);
        }
        // the two poles, which no rhombus edge list contains (L365-388)
        for (int i : {11, 0})
            if (inRange(vertex(i), only_half_sphere, min_z, max_z)) sampling_points_vector.push_back(vertex(i));
        // edges (L391-417)
        for (size_t i = 0; i < edgeStart.size(); i++) {
            const Vec3 &v = (i < number_of_samples * 10 - 15) ? edgeStart[i] : edgeEnd[i];
            if (inRange(v, only_half_sphere, min_z, max_z)) sampling_points_vector.push_back(v);
        }
        // interior of the rhombi: rows between corresponding points of the two edge lists (L446-467)
        int j = 0;
        bool j_flag = false;
        for (size_t i = 0; i < edgeStart.size(); i++) {
            if ((j % (int)(number_of_samples - 1)) == 0 && j != 0) { j = 0; j_flag = true; }
            if ((j % (int)(number_of_samples - 2)) == 0 && j != 0 && j_flag) { j = 0; j_flag = false; }
            fillDistance(edgeStart[i], edgeEnd[i], (j + 1) % (int)number_of_samples, only_half_sphere, min_z, max_z);
            j++;
        }
        // as angles (L514-526)
        for (const Vec3 &v : sampling_points_vector) {
            double rot = std::atan2(v[1], v[0]), tilt = std::acos(v[2]);
            if (tilt < 0.) tilt += M_PI;
            sampling_points_angles.push_back({rot * 180. / M_PI, tilt * 180. / M_PI, 0.});
        }
        numberSamplesAsymmetricUnit = sampling_points_vector.size();
    }

    // asymmetric units (removeRedundantPoints, sampling.cpp:702-1216): a wedge in (rot, tilt) and / or half
    // spaces through the origin. family: "c" "cv" "ch" "s" "d" "dv" "dh" (+ order), "ci" "cs", "t" "td" "th",
    // "o" "oh", "i1".."i4", "i1h".."i4h"
    void removeRedundantPoints(const std::string &group, int order)
    {
        no_redundant_sampling_points_vector.clear();
        no_redundant_sampling_points_angles.clear();
        no_redundant_sampling_points_index.clear();
        auto unit = [](Vec3 v) { const double n = std::sqrt(dot(v, v)); return Vec3{v[0] / n, v[1] / n, v[2] / n}; };
        auto turnY = [](double tiltDeg, const Vec3 &v) {   // Euler_angles2matrix(0, tilt, 0) * v
            const double b = tiltDeg * M_PI / 180., cb = std::cos(b), sb = std::sin(b);
            return Vec3{cb * v[0] - sb * v[2], v[1], sb * v[0] + cb * v[2]};
        };
        std::vector<Vec3> halfSpaces;                       // kept: dot(v, n) >= 0 for every n
        double rotLo = -1e9, rotHi = 1e9, tiltHi = 1e9;     // kept: rotLo <= rot <= rotHi and tilt <= tiltHi ...
        bool orRotZero = false;                             // ... or rot == 0 (the poles; t and o, L820, L903)
        const Vec3 i2a = {0., 1., 0.}, i2b = {-0.4999999839058737, -0.8090170074556163, 0.3090169861701543},
                   i2c = {0.4999999839058737, -0.8090170074556163, 0.3090169861701543};
        const Vec3 i4a = {0., 0., 1.}, i4b = {0.187592467856686, -0.303530987314591, -0.491123477863004},
                   i4c = {0.187592467856686, 0.303530987314591, -0.491123477863004};
        auto neg = [](const Vec3 &v) { return Vec3{-v[0], -v[1], -v[2]}; };
        if (group == "c") { rotLo = -180. / order; rotHi = 180. / order; }
        else if (group == "ci" || group == "cs") tiltHi = 90.;                                  // L714-725
        else if (group == "cv") { rotLo = 0.; rotHi = 180. / order; }                           // L726-737
        else if (group == "ch") { rotLo = -180. / order; rotHi = 180. / order; tiltHi = 90.; }  // L738-751
        else if (group == "s") { rotLo = -360. / order; rotHi = 360. / order; tiltHi = 90.; }   // L752-765
        else if (group == "d") { rotLo = -180. / order + 90.; rotHi = 180. / order + 90.; tiltHi = 90.; }
        else if (group == "dv") { rotLo = 90.; rotHi = 180. / order + 90.; tiltHi = 90.; }      // L780-793
        else if (group == "dh") { rotLo = 0.; rotHi = 180. / order; tiltHi = 90.; }             // L794-807
        else if (group == "t") {
            halfSpaces = {unit({-0.942809, 0., 0.}), unit({0.471405, 0.272165, 0.7698}), unit({0.471404, 0.816497, 0.})};
            rotLo = 90.; rotHi = 150.; orRotZero = true;
        } else if (group == "td")                                                               // L836-863
            halfSpaces = {unit({-0.942809, 0., 0.}), unit({0.471405, 0.272165, 0.7698}), unit({0., 0.471405, -0.666667})};
        else if (group == "th")                                                                 // L864-891
            halfSpaces = {unit({-0.816496, 0., 0.}), unit({0.707107, 0.408248, -0.57735}), unit({-0.408248, -0.707107, 0.})};
        else if (group == "o" || group == "oh") {
            halfSpaces = {unit({0., -1., 1.}), unit({1., 1., 0.}), unit({-1., 1., 0.})};
            if (group == "o") { rotLo = 45.; rotHi = 135.; tiltHi = 90.; orRotZero = true; }
            else { rotLo = 90.; rotHi = 135.; tiltHi = 90.; }                                   // L921-945
        } else if (group == "i1" || group == "i2" || group == "i3") {
            const double tilt = group == "i1" ? 90. : group == "i3" ? 31.7174745559 : 0.;
            halfSpaces = {unit(turnY(tilt, i2a)), unit(turnY(tilt, i2b)), unit(turnY(tilt, i2c))};
        } else if (group == "i4")
            halfSpaces = {neg(unit(turnY(-31.7174745559, i4a))), neg(unit(turnY(-31.7174745559, i4b))), neg(unit(turnY(-31.7174745559, i4c)))};
        else if (group == "i2h" || group == "i1h") {                                            // L1077-1140
            const double tilt = group == "i1h" ? 90. : 0.;
            halfSpaces = {unit(turnY(tilt, i2a)), unit(turnY(tilt, i2b)), unit(turnY(tilt, {1., 0., 0.}))};
        } else if (group == "i3h")                                                              // L1141-1178
            halfSpaces = {unit(turnY(31.7174745559, i4b)), unit(turnY(31.7174745559, i4c)), unit(turnY(31.7174745559, i4a)), {0., 1., 0.}};
        else if (group == "i4h")                                                                // L1179-1215
            halfSpaces = {neg(unit(turnY(-31.7174745559, i4b))), neg(unit(turnY(-31.7174745559, i4c))), neg(unit(turnY(-31.7174745559, i4a))), {0., 1., 0.}};
        else
            REPORT_ERROR(ERR_ARG_INCORRECT, "removeRedundantPoints: unknown point group family '" + group + "'");
        for (size_t i = 0; i < sampling_points_angles.size(); i++) {
            const double rot = sampling_points_angles[i][0], tilt = sampling_points_angles[i][1];
            const Vec3 &v = sampling_points_vector[i];
            bool keep = (rot >= rotLo && rot <= rotHi && tilt <= tiltHi) || (orRotZero && rot == 0.);
            for (const Vec3 &n : halfSpaces) keep = keep && dot(v, n) >= 0;
            if (keep) {
                no_redundant_sampling_points_angles.push_back(sampling_points_angles[i]);
                no_redundant_sampling_points_vector.push_back(sampling_points_vector[i]);
            }
        }
        no_redundant_sampling_points_index.resize(no_redundant_sampling_points_angles.size());
        for (size_t i = 0; i < no_redundant_sampling_points_index.size(); ++i) no_redundant_sampling_points_index[i] = i;
        numberSamplesAsymmetricUnit = no_redundant_sampling_points_vector.size();
    }

    void fillLRRepository(const SymList &SL)
    {
        R_repository.clear();
        R_repository.push_back({1, 0, 0, 0, 1, 0, 0, 0, 1});
        for (const auto &R : SL.R) R_repository.push_back(R);
    }

    // Euler_direction(rot, tilt, psi): third row of the Euler matrix
    static Vec3 eulerDirection(double rot, double tilt)
    {
        const double a = rot * M_PI / 180., b = tilt * M_PI / 180.;
        return {std::sin(b) * std::cos(a), std::sin(b) * std::sin(a), std::cos(b)};
    }
    void fillExpDataProjectionDirectionByLR(const MetaDataVec &DFi)
    {
        exp_data_fileNames.clear();
        exp_data_projection_direction_by_L_R.clear();
        for (size_t id = 0; id < DFi.size(); ++id) {
            const Vec3 d = eulerDirection(DFi.getDouble("angleRot", id, 0), DFi.getDouble("angleTilt", id, 0));
            std::string name;
            DFi.getValue("image", name, id);
            exp_data_fileNames.push_back(name);
            // L * (d^T R)^T with L = I (see SymList: what the reference's i3h fixtures are met with)
            for (const auto &R : R_repository)
                exp_data_projection_direction_by_L_R.push_back({d[0] * R[0] + d[1] * R[3] + d[2] * R[6], d[0] * R[1] + d[1] * R[4] + d[2] * R[7],
                                                                d[0] * R[2] + d[1] * R[5] + d[2] * R[8]});
        }
    }

    void removePointsFarAwayFromExperimentalData()
    {
        if (no_redundant_sampling_points_vector.empty()) return;
        size_t my_end = no_redundant_sampling_points_vector.size() - 1;
        for (size_t i = 0; i <= my_end && my_end != (size_t)-1; i++) {
            bool del = true;
            for (size_t j = 0; del && j < exp_data_projection_direction_by_L_R.size(); j++)
                if (dot(no_redundant_sampling_points_vector[i], exp_data_projection_direction_by_L_R[j]) > cos_neighborhood_radius) del = false;
            if (del) {
                // the last element takes the place of the deleted one, which is examined again
                no_redundant_sampling_points_vector[i] = no_redundant_sampling_points_vector[my_end];
                no_redundant_sampling_points_vector.pop_back();
                no_redundant_sampling_points_angles[i] = no_redundant_sampling_points_angles[my_end];
                no_redundant_sampling_points_angles.pop_back();
                no_redundant_sampling_points_index[i] = no_redundant_sampling_points_index[my_end];
                no_redundant_sampling_points_index.pop_back();
                --my_end;
                --i;
            }
        }
    }

    void computeNeighbors(bool only_winner)
    {
        my_neighbors.clear();
        const size_t nexp = exp_data_projection_direction_by_L_R.size(), nR = R_repository.size();
        const size_t npts = no_redundant_sampling_points_vector.size();
        for (size_t j = 0; j < nexp;) {
            std::vector<size_t> aux;
            if (cos_neighborhood_radius <= -1.0) {
                aux = no_redundant_sampling_points_index;
                j += nR;
            } else {
                for (size_t k = 0; k < nR; k++, j++) {
                    double winner = -1.;
                    for (size_t i = 0; i < npts; ++i) {
                        const double d = dot(no_redundant_sampling_points_vector[i], exp_data_projection_direction_by_L_R[j]);
                        if (!(d > cos_neighborhood_radius)) continue;
                        if (aux.empty()) { aux.push_back(no_redundant_sampling_points_index[i]); winner = d; continue; }
                        bool isNew = true;
                        if (only_winner) {
                            if (winner < d) { if (winner != -1) aux.pop_back(); winner = d; }
                            else isNew = false;
                        } else
                            isNew = std::find(aux.begin(), aux.end(), no_redundant_sampling_points_index[i]) == aux.end();
                        if (isNew) aux.push_back(no_redundant_sampling_points_index[i]);
                    }
                }
            }
            my_neighbors.push_back(aux);
        }
    }

    // findClosestSamplingPoint (sampling.cpp:1991-2098): for every experimental image the direction of the asymmetric unit
    // closest to any of its symmetry mates (first maximum of the dot product) -> <root>_closest_sampling_points.doc.  The
    // reference also stores the image's original angles and shifts as a per-row star comment, which carries no data a reader
    // of the file sees; it is left out.
    void findClosestSamplingPoint(const MetaDataVec &DFi, const std::string &root) const
    {
        MetaDataVec DFo;
        DFo.comment = "Original rot, tilt, psi, Xoff, Yoff are stored as comments";
        const size_t nR = R_repository.size();
        size_t row = 0;
        for (size_t i = 0; i < exp_data_projection_direction_by_L_R.size(); ++row) {
            double best = -2;
            long winner = -1;
            for (size_t k = 0; k < nR; k++, i++)
                for (size_t j = 0; j < no_redundant_sampling_points_vector.size(); j++) {
                    const double d = dot(exp_data_projection_direction_by_L_R[i], no_redundant_sampling_points_vector[j]);
                    if (d > best) { best = d; winner = (long)j; }
                }
            if (winner < 0) REPORT_ERROR(ERR_VALUE_INCORRECT, "findClosestSamplingPoint: no projection directions");
            std::string fnImg;
            DFi.getValue("image", fnImg, row);
            const size_t id = DFo.addObject();
            DFo.setValue("image", fnImg, id);
            DFo.setValue("ref", winner, id);
            DFo.setValue("neighbor", (long)no_redundant_sampling_points_index[winner], id);
            DFo.setValue("angleRot", no_redundant_sampling_points_angles[winner][0], id);
            DFo.setValue("angleTilt", no_redundant_sampling_points_angles[winner][1], id);
            DFo.setValue("anglePsi", no_redundant_sampling_points_angles[winner][2], id);
        }
        if (!root.empty()) DFo.write(root + "_closest_sampling_points.doc");
    }

    void createAsymUnitFile(const std::string &root) const
    {
        MetaDataVec DF;
        for (size_t i = 0; i < no_redundant_sampling_points_vector.size(); i++) {
            const size_t id = DF.addObject();
            DF.setValue("ref", (long)i, id);
            DF.setValue("neighbor", (long)no_redundant_sampling_points_index[i], id);
            DF.setValue("angleRot", no_redundant_sampling_points_angles[i][0], id);
            DF.setValue("angleTilt", no_redundant_sampling_points_angles[i][1], id);
            DF.setValue("anglePsi", no_redundant_sampling_points_angles[i][2], id);
            DF.setValue("X", no_redundant_sampling_points_vector[i][0], id);
            DF.setValue("Y", no_redundant_sampling_points_vector[i][1], id);
            DF.setValue("Z", no_redundant_sampling_points_vector[i][2], id);
        }
        DF.comment = "REF refers to the projection directions BEFORE delete those not in a neighborhood, ";
        DF.write(root + "_angles.doc");
    }

    void saveSamplingFile(const std::string &root, bool write_vectors) const
    {
        const std::string fn = root + "_sampling.xmd";
        {
            MetaDataVec md;
            const size_t id = md.addObject();
            md.setValue("sampling_rate", sampling_rate_rad, id);
            md.setValue("neighborhoodRadius", cos_neighborhood_radius, id);
            md.setValue("pointsAsymmetricUnit", (long)numberSamplesAsymmetricUnit, id);
            md.comment = "data_extra -> sampling description; data_neighbors --> List with order of eachexperimental images and its neighbors";
            md.write("extra@" + fn, false);
        }
        {
            MetaDataVec md;
            md.addLabel("neighbor");
            if (!exp_data_fileNames.empty()) md.addLabel("image");
            md.addLabel("neighbors");
            for (size_t i = 0; i < my_neighbors.size(); ++i) {
                const size_t id = md.addObject();
                md.setValue("neighbor", (long)(i + 1), id);
                if (!exp_data_fileNames.empty()) md.setValue("image", exp_data_fileNames[i], id);
                std::string s = " ";
                for (size_t v : my_neighbors[i]) s += std::to_string(v) + " ";
                md.setValue("neighbors", s, id);
            }
            md.write("neighbors@" + fn, true);
        }
        {
            MetaDataVec md;
            for (const char *l : {"neighbor", "angleRot", "angleTilt", "anglePsi"}) md.addLabel(l);
            if (write_vectors) for (const char *l : {"X", "Y", "Z"}) md.addLabel(l);
            for (size_t i = 0; i < no_redundant_sampling_points_index.size(); ++i) {
                const size_t id = md.addObject();
                md.setValue("neighbor", (long)no_redundant_sampling_points_index[i], id);
                md.setValue("angleRot", no_redundant_sampling_points_angles[i][0], id);
                md.setValue("angleTilt", no_redundant_sampling_points_angles[i][1], id);
                md.setValue("anglePsi", no_redundant_sampling_points_angles[i][2], id);
                if (write_vectors) {
                    md.setValue("X", no_redundant_sampling_points_vector[i][0], id);
                    md.setValue("Y", no_redundant_sampling_points_vector[i][1], id);
                    md.setValue("Z", no_redundant_sampling_points_vector[i][2], id);
                }
            }
            md.write("projectionDirections@" + fn, true);
        }
    }
};

}  // namespace mc
#endif
