// The three matrices of Winograd F(4x4,3x3) in sivo_amd/csrc/wino4_transforms.hpp (round 6: interpolation points 0, 1, -1, 1/2, -2, inf),
// compiled for the HOST with hipcc: Y = A^T [(G g G^T) .* (B^T d B)] A equals the 4 x 4 valid correlation of a 6 x 6 patch with a 3 x 3
// filter, and B^T d B is EXACT in fp32 on small integers (every entry of B^T is a dyadic rational).  tests/test_wino4_transforms_cpu.py runs it.
#include <cmath>
#include <cstdio>
#include <cstdlib>

#include "../../sivo_amd/csrc/wino4_transforms.hpp"

int main() {
    using namespace sivo;
    static const double BT[6][6] = {{1, -1.5, -2, 1.5, 1, 0}, {0, -1, 0.5, 2.5, 1, 0}, {0, 1, -2.5, 0.5, 1, 0},
                                    {0, -2, -1, 2, 1, 0},     {0, 0.5, -1, -0.5, 1, 0}, {0, 1, -1.5, -2, 1.5, 1}};
    unsigned seed = 12345u;
    auto rnd = [&]() { seed = seed * 1664525u + 1013904223u; return (int)((seed >> 16) % 17) - 8; };
    double worst = 0;
    for (int trial = 0; trial < 2000; ++trial) {
        float d[6][6], g[3][3];
        for (auto &r : d) for (float &v : r) v = (float)rnd();
        for (auto &r : g) for (float &v : r) v = (float)rnd();
        // input transform: columns, then rows (as the kernels do), against the matrix in f64: exact
        float t[6][6], V[6][6];
        for (int j = 0; j < 6; ++j) { float c[6]; wino4_bt(d[0][j], d[1][j], d[2][j], d[3][j], d[4][j], d[5][j], c); for (int i = 0; i < 6; ++i) t[i][j] = c[i]; }
        for (int i = 0; i < 6; ++i) wino4_bt(t[i][0], t[i][1], t[i][2], t[i][3], t[i][4], t[i][5], V[i]);
        for (int i = 0; i < 6; ++i)
            for (int j = 0; j < 6; ++j) {
                double ref = 0;
                for (int a = 0; a < 6; ++a) for (int b = 0; b < 6; ++b) ref += BT[i][a] * d[a][b] * BT[j][b];
                if ((double)V[i][j] != ref) { std::printf("B^T d B not exact at (%d, %d): %g vs %g\n", i, j, (double)V[i][j], ref); return 1; }
            }
        // weights in f64, products in f64, output transform in fp32
        double U[6][6];
        for (int i = 0; i < 6; ++i)
            for (int j = 0; j < 6; ++j) {
                double s = 0;
                for (int a = 0; a < 3; ++a) for (int b = 0; b < 3; ++b) s += WINO4_G[i][a] * g[a][b] * WINO4_G[j][b];
                U[i][j] = s;
            }
        float m[6][6], tt[4][6], Y[4][4];
        for (int i = 0; i < 6; ++i) for (int j = 0; j < 6; ++j) m[i][j] = (float)(U[i][j] * V[i][j]);
        for (int j = 0; j < 6; ++j) { float s4[4]; wino4_at(m[0][j], m[1][j], m[2][j], m[3][j], m[4][j], m[5][j], s4); for (int i = 0; i < 4; ++i) tt[i][j] = s4[i]; }
        for (int i = 0; i < 4; ++i) wino4_at(tt[i][0], tt[i][1], tt[i][2], tt[i][3], tt[i][4], tt[i][5], Y[i]);
        for (int i = 0; i < 4; ++i)
            for (int j = 0; j < 4; ++j) {
                double ref = 0;
                for (int a = 0; a < 3; ++a) for (int b = 0; b < 3; ++b) ref += (double)d[i + a][j + b] * g[a][b];
                worst = std::fmax(worst, std::fabs((double)Y[i][j] - ref));
            }
    }
    std::printf("worst |Y - correlation| over 2000 integer patches: %.3g\n", worst);
    return worst < 2e-3 ? 0 : 2;      // (|values| up to ~5000 pass through fp32 in M and the output transform)
}
