// kitti_io.hpp — the two file formats at the ends of SIVO's frame loop.
//   loadImages:          reference src/sivo.cc:145-177 — <sequence>/times.txt (one timestamp per line, blank
//                        lines skipped) and the image_2 / image_3 file names "%06d.png".
//   saveTrajectoryKITTI: the line format of System::SaveTrajectoryKITTI (src/orbslam/System.cc:322-329):
//                        per frame the 3x4 [Rwc | twc] of the camera-to-world pose, row-major, `fixed`,
//                        setprecision(9), single-space separated.  The keyframe-relative bookkeeping that
//                        produces Tcw (System.cc:300-320) is SLAM state and stays with the caller.
#pragma once
#include <fstream>
#include <iomanip>
#include <sstream>
#include <string>
#include <vector>

namespace SIVO {

inline void loadImages(const std::string &strPathToSequence, std::vector<std::string> &vstrImageLeft,
                       std::vector<std::string> &vstrImageRight, std::vector<double> &vTimestamps) {
    std::ifstream fTimes((strPathToSequence + "/times.txt").c_str());
    std::string line;
    while (std::getline(fTimes, line)) {
        if (line.empty()) continue;
        std::stringstream ss(line);
        double t;
        ss >> t;
        vTimestamps.push_back(t);
    }
    const size_t n = vTimestamps.size();
    vstrImageLeft.resize(n);
    vstrImageRight.resize(n);
    for (size_t i = 0; i < n; ++i) {
        std::stringstream ss;
        ss << std::setfill('0') << std::setw(6) << i;
        vstrImageLeft[i] = strPathToSequence + "/image_2/" + ss.str() + ".png";
        vstrImageRight[i] = strPathToSequence + "/image_3/" + ss.str() + ".png";
    }
}

// Tcw: 12 floats per frame (Rcw row-major, tcw), already expressed relative to the first keyframe.
inline bool saveTrajectoryKITTI(const std::string &filename, const std::vector<float> &Tcw) {
    std::ofstream f(filename.c_str());
    if (!f) return false;
    f << std::fixed;
    for (size_t k = 0; k + 12 <= Tcw.size(); k += 12) {
        const float *R = &Tcw[k], *t = R + 9;
        float twc[3];
        for (int i = 0; i < 3; ++i) twc[i] = -(R[i] * t[0] + R[3 + i] * t[1] + R[6 + i] * t[2]);   // -Rwc * tcw, Rwc = Rcw'
        f << std::setprecision(9);
        for (int i = 0; i < 3; ++i) {
            f << R[i] << " " << R[3 + i] << " " << R[6 + i] << " " << twc[i];
            f << (i < 2 ? " " : "\n");
        }
    }
    return true;
}

}  // namespace SIVO
