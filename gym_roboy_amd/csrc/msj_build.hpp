// msj_build.hpp - fold a generic robot description (rb_robot_desc) into the
// constants of the ball-joint closed form (msj_math.hpp), or say why the
// robot is not of that class.  Host code, fp64 until the final cast.
#pragma once
#include <cmath>
#include <string>

#include "../../include/roboy_sim.h"
#include "msj_math.hpp"

namespace rb {

// true when the description is "one body on an x-y-z ball joint at the base
// origin, via-points only on the base and on that body"
inline bool msj_class(const rb_robot_desc *d, std::string &why) {
    if (d->n_q != 3) { why = "n_q != 3"; return false; }
    for (int i = 0; i < 3; ++i) {
        if (d->parent[i] != i - 1) { why = "joints are not a serial chain"; return false; }
        for (int a = 0; a < 3; ++a) {
            if (d->origin[3 * i + a] != 0.0) { why = "joint origins are not co-located at the base origin"; return false; }
            if (std::fabs(d->axis[3 * i + a] - (a == i ? 1.0 : 0.0)) > 1e-12) { why = "joint axes are not x, y, z"; return false; }
        }
    }
    for (int i = 0; i < 2; ++i) {
        if (d->mass[i] != 0.0) { why = "virtual links 0/1 carry mass"; return false; }
        for (int a = 0; a < 6; ++a)
            if (d->inertia[6 * i + a] != 0.0) { why = "virtual links 0/1 carry inertia"; return false; }
    }
    for (int k = 0; k < d->n_t; ++k) {
        const int v0 = d->vp_offset[k], v1 = d->vp_offset[k + 1];
        if (v1 - v0 < 2) { why = "tendon with fewer than two via-points"; return false; }
        bool on_body = false;
        for (int v = v0; v < v1; ++v) {
            const int link = d->vp_link[v];
            if (link != -1 && link != 2) { why = "via-point on a virtual link"; return false; }
            if (link == 2) on_body = true;
            else if (on_body) { why = "tendon returns from the body to the base"; return false; }
        }
        if (d->vp_link[v0] != -1 || !on_body) { why = "tendon does not run base -> body"; return false; }
    }
    return true;
}

template <typename T, int NT>
int msj_build(const rb_robot_desc *d, double step_size, int nsub, MsjConst<T, NT> *out, std::string &err,
              bool exact = true) {
    if (!msj_class(d, err)) return RB_EUNSUPPORTED;
    // exact = true: the kernels with a compile-time tendon count; false: up to NT tendons, count in c.nt
    if (exact ? d->n_t != NT : (d->n_t < 1 || d->n_t > NT)) {
        err = "tendon count does not match this kernel instance"; return RB_EUNSUPPORTED;
    }
    MsjConst<T, NT> &c = *out;
    const double log2e = 1.4426950408889634;
    const double sc = std::sqrt(log2e) / d->fl_width;     // strain scale: f_L = exp2(-(sc e)^2)
    auto seglen = [&](int va, int vb) {
        double s = 0.0;
        for (int a = 0; a < 3; ++a) { const double t = d->vp_pos[3 * vb + a] - d->vp_pos[3 * va + a]; s += t * t; }
        return std::sqrt(s);
    };
    for (int k = 0; k < d->n_t; ++k) {
        const int v0 = d->vp_offset[k], v1 = d->vp_offset[k + 1];
        int vb = v0;
        while (d->vp_link[vb] == -1) ++vb;   // first body via-point
        const int va = vb - 1;               // last base via-point
        double lc = 0.0;
        for (int v = v0; v + 1 < v1; ++v)
            if (v != va) lc += seglen(v, v + 1);
        const double moving0 = seglen(va, vb);   // R = I in the zero pose
        if (moving0 < 1e-6) { err = "degenerate tendon segment"; return RB_EINVAL; }
        const double l0 = lc + moving0;
        MsjTendon<T> &t = c.ten[k];
        const double vl0 = d->v_max * l0;
        for (int a = 0; a < 3; ++a) { t.A[a] = T(d->vp_pos[3 * va + a]); t.Bv[a] = T(d->vp_pos[3 * vb + a] / vl0); }
        double ab2 = 0.0;
        for (int a = 0; a < 3; ++a) ab2 += d->vp_pos[3 * va + a] * d->vp_pos[3 * va + a] + d->vp_pos[3 * vb + a] * d->vp_pos[3 * vb + a];
        for (int a = 0; a < 3; ++a) t.B2[a] = T(-2.0 * d->vp_pos[3 * vb + a]);
        t.ab2 = T(ab2); t.il0s = T(sc / l0); t.elcs = T(sc * (lc / l0 - 1.0)); t.ksg = T(d->kp * d->setpoint_scale / l0);
        t.fmaxv = T(d->f_max[k] * vl0);
        t.pad[0] = t.pad[1] = T(0);
    }
    for (int k = d->n_t; k < NT; ++k) {       // unused records: a well-formed tendon that never pulls
        MsjTendon<T> &t = c.ten[k];
        t.A[0] = T(1); t.A[1] = T(0); t.A[2] = T(0); t.Bv[0] = T(0); t.Bv[1] = T(0); t.Bv[2] = T(1);
        t.B2[0] = T(0); t.B2[1] = T(0); t.B2[2] = T(-2);
        t.ab2 = T(2); t.il0s = T(sc); t.elcs = T(-sc); t.ksg = T(0); t.fmaxv = T(0);
        t.pad[0] = t.pad[1] = T(0);
    }
    c.nt = d->n_t;
    const double m = d->mass[2];
    const double *cm = d->com + 6, *ic = d->inertia + 12;
    const double c2 = cm[0] * cm[0] + cm[1] * cm[1] + cm[2] * cm[2];
    c.IO[0] = T(ic[0] + m * (c2 - cm[0] * cm[0]));
    c.IO[1] = T(ic[1] + m * (c2 - cm[1] * cm[1]));
    c.IO[2] = T(ic[2] + m * (c2 - cm[2] * cm[2]));
    c.IO[3] = T(ic[3] - m * cm[0] * cm[1]);
    c.IO[4] = T(ic[4] - m * cm[0] * cm[2]);
    c.IO[5] = T(ic[5] - m * cm[1] * cm[2]);
    for (int a = 0; a < 3; ++a) {
        c.mc[a] = T(m * cm[a]);
        c.g[a] = T(d->gravity[a]);
        c.arm[a] = T(d->armature[a]); c.damp[a] = T(d->damping[a]);
        c.qlo[a] = T(d->q_lo[a]); c.qhi[a] = T(d->q_hi[a]); c.qdmax[a] = T(d->qd_max[a]);
    }
    c.kps = T(d->kp / sc);
    c.pe_k2s = T(log2e * d->kpe / (d->e0 * sc));
    c.inv_pe_den = T(1.0 / (std::exp(d->kpe) - 1.0));
    const double slope0 = 1.0 + 1.0 / d->fv_a;
    const double c2l = slope0 / (d->fv_n - 1.0);
    c.fv_c2s = T(-1.0 / d->fv_a);
    c.fv_k = T(1.0 + 1.0 / d->fv_a);
    c.fv_c1l = T(d->fv_n * c2l); c.fv_c2l = T(c2l);
    c.h = T(step_size / nsub); c.nsub = nsub;
    c.simple = (ic[3] == 0.0 && ic[4] == 0.0 && ic[5] == 0.0 && cm[0] == 0.0 && cm[1] == 0.0 &&
                d->gravity[0] == 0.0 && d->gravity[1] == 0.0) ? 1 : 0;
    return RB_OK;
}

// Mirror symmetry of an 8-tendon ball-joint robot (msj_kernels.hpp, "mirror pairs"): a reflection S = diag(1,-1,1) (x-z
// plane, mirror = 0) or diag(-1,1,1) (y-z plane, mirror = 1) that maps the tendon set onto itself without fixed points -
// A' = S A, B' = S B bit for bit, same muscle constants - for a body that is itself symmetric (c.simple: principal-axis
// inertia, centre of mass on z, gravity along z) with symmetric limits on the two joints the reflection turns round.
// half[k] / image[k]: one tendon of each pair (the lower index, ascending) and its image.
template <typename T>
bool find_mirror_pairs(const MsjConst<T, 8> &c, int &mirror, int half[4], int image[4]) {
    if (!c.simple || c.nt != 8) return false;
    for (int m = 0; m < 2; ++m) {
        const int ax = m == 0 ? 1 : 0;                      // the coordinate the reflection negates
        const int f0 = m == 0 ? 0 : 1, f1 = 2;              // joints whose angle changes sign in the mirror
        if (c.qlo[f0] != -c.qhi[f0] || c.qlo[f1] != -c.qhi[f1]) continue;
        int partner[8];
        bool ok = true;
        for (int k = 0; k < 8 && ok; ++k) {
            partner[k] = -1;
            for (int j = 0; j < 8 && partner[k] < 0; ++j) {
                const MsjTendon<T> &a = c.ten[k], &b = c.ten[j];
                bool img = j != k && a.ab2 == b.ab2 && a.il0s == b.il0s && a.elcs == b.elcs && a.ksg == b.ksg && a.fmaxv == b.fmaxv;
                for (int e = 0; e < 3 && img; ++e) {
                    const T sg = e == ax ? T(-1) : T(1);
                    img = b.A[e] == sg * a.A[e] && b.Bv[e] == sg * a.Bv[e] && b.B2[e] == sg * a.B2[e];
                }
                if (img) partner[k] = j;
            }
            ok = partner[k] >= 0;
        }
        for (int k = 0; k < 8 && ok; ++k) ok = partner[partner[k]] == k;      // an involution
        if (!ok) continue;
        int n = 0;
        for (int k = 0; k < 8; ++k)
            if (k < partner[k]) { if (n < 4) { half[n] = k; image[n] = partner[k]; } ++n; }
        if (n != 4) continue;
        mirror = m;
        return true;
    }
    return false;
}

}  // namespace rb
