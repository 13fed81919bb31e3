/*
 * oracle/match_oracle.c — TEST INFRASTRUCTURE ONLY (never linked into the product).
 *
 * CPU restatement of the Hamming-matching part of navganti/SIVO's per-frame
 * path:
 *   - ORBmatcher::DescriptorDistance          src/orbslam/ORBmatcher.cc:1582-1596
 *   - the candidate-list brute-force argmin with best / second-best used by
 *     every Search* routine (e.g. SearchByProjection, ORBmatcher.cc:44-127,
 *     best/second update at :86-104)
 *   - Frame::ComputeStereoMatches             src/orbslam/Frame.cc:444-629
 *
 * PARITY PINNED AGAINST THE REFERENCE'S OWN CODE: orc_descriptor_distance and the best / second-best scan against
 * ORBmatcher.cc (tests/cpp/pin_matcher.cpp, tests/test_pin_matcher.py); orc_stereo_matches against Frame.cc's
 * ComputeStereoMatches, both compiled untouched into oracle/_ref: mvRight / mvDepth bit for bit on 6 stereo scenes
 * (tests/test_pin_frame.py; fixture tests/golden/frame_reference.json).
 */
#include <limits.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define TH_HIGH 100
#define TH_LOW 50

/* ORBmatcher.cc:1582-1596 — 8 x (xor, SWAR popcount) over 32-bit words. */
int orc_descriptor_distance(const uint8_t *a, const uint8_t *b) {
    int dist = 0;
    for (int i = 0; i < 8; ++i) {
        uint32_t pa, pb;
        memcpy(&pa, a + 4 * i, 4);
        memcpy(&pb, b + 4 * i, 4);
        uint32_t v = pa ^ pb;
        v = v - ((v >> 1) & 0x55555555u);
        v = (v & 0x33333333u) + ((v >> 2) & 0x33333333u);
        dist += (int)((((v + (v >> 4)) & 0xF0F0F0Fu) * 0x1010101u) >> 24);
    }
    return dist;
}

/* Dense nA x nB distance matrix (the MapPoint::ComputeDistinctiveDescriptors
 * pattern, src/orbslam/MapPoint.cc:284-347, and the brute-force reference for
 * every candidate search). */
void orc_hamming_matrix(const uint8_t *A, int nA, const uint8_t *B, int nB, int32_t *out) {
#pragma omp parallel for schedule(static)
    for (int i = 0; i < nA; ++i)
        for (int j = 0; j < nB; ++j)
            out[(int64_t)i * nB + j] = orc_descriptor_distance(A + 32 * (int64_t)i, B + 32 * (int64_t)j);
}

/* Candidate-list argmin with best and second best, the inner loop shared by
 * the Search* routines (ORBmatcher.cc:78-104): candidates are visited in list
 * order, `dist < best` moves best to second, `else if dist < second` updates
 * second.  Initial values are 256 (ORBmatcher.cc:73-76).  cand_idx[cand_off[i]
 * .. cand_off[i+1]) are row indices into B for query i. */
void orc_hamming_argmin2(const uint8_t *A, int nA, const uint8_t *B, const int32_t *cand_off,
                         const int32_t *cand_idx, int32_t *best_idx, int32_t *best_dist,
                         int32_t *second_dist) {
    for (int i = 0; i < nA; ++i) {
        int best = 256, second = 256, bi = -1;
        for (int c = cand_off[i]; c < cand_off[i + 1]; ++c) {
            const int j = cand_idx[c];
            const int d = orc_descriptor_distance(A + 32 * (int64_t)i, B + 32 * (int64_t)j);
            if (d < best) { second = best; best = d; bi = j; }
            else if (d < second) { second = d; }
        }
        best_idx[i] = bi; best_dist[i] = best; second_dist[i] = second;
    }
}

typedef struct { const uint8_t *data; int32_t rows, cols, step; } OrcImage;

/* Frame::ComputeStereoMatches (Frame.cc:444-629).  keys are (x, y, octave)
 * triples as float/float/int in separate arrays.  pyrL/pyrR point at the
 * level images WITHOUT border (the ROI mvImagePyramid[l] exposes).  Outputs
 * uRight/depth are -1 where no match.  Returns the number of matches kept.
 * best_r (optional) receives the Hamming-stage best right index (or -1). */
static int cmp_pair(const void *a, const void *b) {
    const int *p = (const int *)a, *q = (const int *)b;
    if (p[0] != q[0]) return p[0] < q[0] ? -1 : 1;
    return p[1] < q[1] ? -1 : (p[1] > q[1]);
}

int orc_stereo_matches(int nL, const float *lx, const float *ly, const int32_t *loct, const uint8_t *ldesc,
                       int nR, const float *rx, const float *ry, const int32_t *roct, const uint8_t *rdesc,
                       int nlevels, const float *scale, const float *inv_scale,
                       const OrcImage *pyrL, const OrcImage *pyrR, float bf, float b,
                       float *uRight, float *depth, int32_t *best_r) {
    const int thOrbDist = (TH_HIGH + TH_LOW) / 2;              /* :448 */
    const int nRows = pyrL[0].rows;                             /* :450-451 */
    (void)nlevels;

    /* :454-477 row table.  The reference indexes vRowIndices[yi] unchecked
     * (upstream out-of-range hazard); rows are clamped here. */
    int *cnt = (int *)calloc((size_t)nRows + 1, sizeof(int));
    for (int iR = 0; iR < nR; ++iR) {
        const float r = 2.0f * scale[roct[iR]];
        const int maxr = (int)ceilf(ry[iR] + r), minr = (int)floorf(ry[iR] - r);
        for (int yi = minr; yi <= maxr; ++yi)
            if (yi >= 0 && yi < nRows) cnt[yi + 1]++;
    }
    for (int i = 0; i < nRows; ++i) cnt[i + 1] += cnt[i];
    int *rows = (int *)malloc(sizeof(int) * (size_t)(cnt[nRows] > 0 ? cnt[nRows] : 1));
    int *fill = (int *)calloc((size_t)nRows, sizeof(int));
    for (int iR = 0; iR < nR; ++iR) {
        const float r = 2.0f * scale[roct[iR]];
        const int maxr = (int)ceilf(ry[iR] + r), minr = (int)floorf(ry[iR] - r);
        for (int yi = minr; yi <= maxr; ++yi)
            if (yi >= 0 && yi < nRows) rows[cnt[yi] + fill[yi]++] = iR;
    }

    const float minZ = b, minD = 0, maxD = bf / minZ;           /* :479-482 */
    int *vDistIdx = (int *)malloc(sizeof(int) * 2 * (size_t)(nL > 0 ? nL : 1));
    int nDist = 0;

    for (int iL = 0; iL < nL; ++iL) { uRight[iL] = -1.0f; depth[iL] = -1.0f; if (best_r) best_r[iL] = -1; }

    for (int iL = 0; iL < nL; ++iL) {
        const int levelL = loct[iL];
        const float vL = ly[iL], uL = lx[iL];
        const int row = (int)vL;                                /* vRowIndices[vL], :496 */
        if (row < 0 || row >= nRows) continue;
        const int c0 = cnt[row], c1 = cnt[row + 1];
        if (c0 == c1) continue;                                 /* :498-500 */
        const float minU = uL - maxD, maxU = uL - minD;
        if (maxU < 0) continue;                                 /* :505-507 */

        int bestDist = TH_HIGH;
        int bestIdxR = 0;
        const uint8_t *dL = ldesc + 32 * (int64_t)iL;
        for (int c = c0; c < c1; ++c) {                         /* :515-535 */
            const int iR = rows[c];
            if (roct[iR] < levelL - 1 || roct[iR] > levelL + 1) continue;
            const float uR = rx[iR];
            if (uR >= minU && uR <= maxU) {
                const int dist = orc_descriptor_distance(dL, rdesc + 32 * (int64_t)iR);
                if (dist < bestDist) { bestDist = dist; bestIdxR = iR; }
            }
        }
        if (best_r && bestDist < TH_HIGH) best_r[iL] = bestIdxR;

        if (bestDist < thOrbDist) {                             /* :538 */
            const float uR0 = rx[bestIdxR];
            const float sf = inv_scale[levelL];
            const float scaleduL = roundf(lx[iL] * sf);
            const float scaledvL = roundf(ly[iL] * sf);
            const float scaleduR0 = roundf(uR0 * sf);
            const int w = 5, L = 5;
            const OrcImage *imL = &pyrL[levelL], *imR = &pyrR[levelL];
            const int cy = (int)scaledvL, cxl = (int)scaleduL;
            float vDists[11];
            int bestD = INT_MAX, bestincR = 0;

            const float iniu = scaleduR0 + L - w, endu = scaleduR0 + L + w + 1;   /* :560-565 */
            if (iniu < 0 || endu >= imR->cols) continue;

            const float cL = (float)imL->data[cy * imL->step + cxl];
            for (int incR = -L; incR <= L; ++incR) {            /* :567-583 */
                const int cxr = (int)scaleduR0 + incR;
                const float cR = (float)imR->data[cy * imR->step + cxr];
                double acc = 0.0;
                for (int dy = -w; dy <= w; ++dy)
                    for (int dx = -w; dx <= w; ++dx) {
                        const float a = (float)imL->data[(cy + dy) * imL->step + cxl + dx] - cL;
                        const float bb = (float)imR->data[(cy + dy) * imR->step + cxr + dx] - cR;
                        acc += fabs((double)(a - bb));
                    }
                const float dist = (float)acc;
                if (dist < (float)bestD) { bestD = (int)dist; bestincR = incR; }
                vDists[L + incR] = dist;
            }
            if (bestincR == -L || bestincR == L) continue;      /* :585-586 */

            const float dist1 = vDists[L + bestincR - 1];
            const float dist2 = vDists[L + bestincR];
            const float dist3 = vDists[L + bestincR + 1];
            const float deltaR = (dist1 - dist3) / (2.0f * (dist1 + dist3 - 2.0f * dist2));
            if (deltaR < -1 || deltaR > 1) continue;            /* :596-597 */

            float bestuR = scale[levelL] * ((float)scaleduR0 + (float)bestincR + deltaR);
            float disparity = uL - bestuR;
            if (disparity >= minD && disparity < maxD) {        /* :605-613 */
                if (disparity <= 0) { disparity = 0.01f; bestuR = (float)(uL - 0.01); }
                depth[iL] = bf / disparity;
                uRight[iL] = bestuR;
                vDistIdx[2 * nDist] = bestD; vDistIdx[2 * nDist + 1] = iL; ++nDist;
            }
        }
    }

    int kept = nDist;
    if (nDist > 0) {                                            /* :617-628 */
        qsort(vDistIdx, (size_t)nDist, 2 * sizeof(int), cmp_pair);
        const float median = (float)vDistIdx[2 * (nDist / 2)];
        const float thDist = 1.5f * 1.4f * median;
        for (int i = nDist - 1; i >= 0; --i) {
            if ((float)vDistIdx[2 * i] < thDist) break;
            uRight[vDistIdx[2 * i + 1]] = -1; depth[vDistIdx[2 * i + 1]] = -1; --kept;
        }
    }
    free(cnt); free(rows); free(fill); free(vDistIdx);
    return kept;
}
