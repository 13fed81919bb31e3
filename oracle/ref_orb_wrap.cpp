// C entry point over the reference's OWN SIVO::ORBextractor, compiled from /root/reference/src/orbslam/ORBextractor.cc
// into oracle/_ref/libref_orb.so (oracle/Makefile `ref`).  OpenCV's primitives under it are the restatements of
// oracle/orb_oracle.c (ref_shims/opencv2/imgproc/imgproc.hpp); everything else is the reference's code.
// Test infrastructure: tests/test_pin_orb.py compares the CPU oracle and the device extractor with it.
//
// The allocator.  ORBextractor::DistributeOctTree sorts std::pair<int, ExtractorNode *> (ORBextractor.cc:675): among
// nodes that hold the same number of keys it expands the one at the HIGHER HEAP ADDRESS first, so which keys survive
// depends on the allocator.  The oracle and the device fix the one order that does not: a node created later compares
// greater.  To compare like with like, `mode 0` runs the reference on an arena that never reuses memory (addresses grow
// with creation order: inside this library only — the version script keeps operator new local); `mode 1` leaves it on
// the process's malloc, to show what that freedom amounts to.
#include <cstdint>
#include <cstring>
#include <vector>

#include "include/orbslam/ORBextractor.h"   // the reference's header, found through -I/root/reference
#include "ref_arena.inc"

// (*mpORBextractorLeft)(im, cv::Mat(), mvKeys, mDescriptors) on a fresh extractor: keys as cv::KeyPoint (28 bytes each),
// descriptors n x 32, then mvImagePyramid (the views without the border, concatenated) and the four scale tables.
extern "C" int ref_orb_extract(int nfeatures, float scaleFactor, int nlevels, int iniThFAST, int minThFAST, int mode, const uint8_t *gray,
                               int rows, int cols, int step, void *keys_out, uint8_t *desc_out, int capacity, uint8_t *levels_out,
                               int levels_capacity, int32_t *level_rows, int32_t *level_cols, float *tables) {
    arena_begin(mode == 0);
    int n = 0;
    {
        SIVO::ORBextractor ex(nfeatures, scaleFactor, nlevels, iniThFAST, minThFAST);
        cv::Mat image(rows, cols, CV_8UC1, const_cast<uint8_t *>(gray), (size_t)step);
        std::vector<cv::KeyPoint> keys;
        cv::Mat desc;
        ex(image, cv::Mat(), keys, desc);
        n = (int)keys.size();
        static_assert(sizeof(cv::KeyPoint) == 28, "cv::KeyPoint layout");
        for (int i = 0; i < n && i < capacity; ++i) {
            std::memcpy(static_cast<uint8_t *>(keys_out) + 28 * (size_t)i, &keys[(size_t)i], 28);
            std::memcpy(desc_out + 32 * (size_t)i, desc.ptr(i), 32);
        }
        size_t off = 0;
        for (int l = 0; l < nlevels; ++l) {
            const cv::Mat &m = ex.mvImagePyramid[(size_t)l];
            level_rows[l] = m.rows; level_cols[l] = m.cols;
            if (off + (size_t)m.rows * m.cols > (size_t)levels_capacity) { n = -1; break; }
            for (int r = 0; r < m.rows; ++r) std::memcpy(levels_out + off + (size_t)r * m.cols, m.ptr(r), (size_t)m.cols);
            off += (size_t)m.rows * m.cols;
        }
        const std::vector<float> a = ex.GetScaleFactors(), b = ex.GetInverseScaleFactors(), c = ex.GetScaleSigmaSquares(),
                                 d = ex.GetInverseScaleSigmaSquares();
        for (int l = 0; l < nlevels; ++l) { tables[l] = a[(size_t)l]; tables[nlevels + l] = b[(size_t)l]; tables[2 * nlevels + l] = c[(size_t)l]; tables[3 * nlevels + l] = d[(size_t)l]; }
    }
    arena_end();
    return n;
}
