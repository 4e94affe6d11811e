// host side of msj_mfma.hpp (was in csrc/msj_build.hpp while the experiment ran)
#pragma once
#include "msj_math.hpp"
namespace rb {
// A operand of the matrix-core form of the tendon routing (msj_mfma.hpp).  With x = (R row-major, 1) in R^10,
// every tendon's d2 = |B - R^T A|^2 and m = (R^T A) x Bv are linear in x:
//   d2   = ab2 + sum_jc R_jc A_j B2_c
//   m_x  = sum_j A_j (R_j1 Bv_z - R_j2 Bv_y),  m_y, m_z cyclic
// 32 rows (tendon k: rows 4k .. 4k+3 = d2, m_x, m_y, m_z) x 10 columns, laid out as v_mfma_f32_32x32x2_f32
// reads its A operand: MFMA p of 5 takes from lane l the element [row l & 31][column 2p + (l >> 5)].
inline void msj_geom_table(const MsjConst<float, 8> &c, float out[5][64]) {
    double G[32][10] = {};
    for (int k = 0; k < 8; ++k) {
        const MsjTendon<float> &t = c.ten[k];
        for (int j = 0; j < 3; ++j) {
            const double A = t.A[j];
            for (int cc = 0; cc < 3; ++cc) G[4 * k][3 * j + cc] = A * t.B2[cc];
            G[4 * k + 1][3 * j + 1] = A * t.Bv[2];  G[4 * k + 1][3 * j + 2] = -A * t.Bv[1];
            G[4 * k + 2][3 * j + 2] = A * t.Bv[0];  G[4 * k + 2][3 * j + 0] = -A * t.Bv[2];
            G[4 * k + 3][3 * j + 0] = A * t.Bv[1];  G[4 * k + 3][3 * j + 1] = -A * t.Bv[0];
        }
        G[4 * k][9] = t.ab2;
    }
    for (int p = 0; p < 5; ++p)
        for (int l = 0; l < 64; ++l) out[p][l] = float(G[l & 31][2 * p + (l >> 5)]);
}

}  // namespace rb
