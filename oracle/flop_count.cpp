// TEST / MEASUREMENT INFRASTRUCTURE - instrumented restatement of the physics step:
// counts the floating-point operations of one env step (SURVEY.md §8(d),
// "Algorithmic flops ... emit the exact count from an instrumented oracle").
//
// The closed form of the ball-joint class (gym_roboy_amd/csrc/msj_math.hpp: the
// arithmetic the env-per-lane kernels execute) is a template over its scalar
// type; here it is instantiated with a scalar that tallies every operation it
// takes part in.  The values are carried along in fp64, so the same run also
// reproduces the step (tests/test_flop_count.py checks it against the C oracle:
// a count taken on a wrong trajectory would be worthless).
//
// What is counted, per env step, along the path the given (state, action) takes
// (the arithmetic is branch-free apart from the wave-uniform `simple` switch, so
// the count does not depend on the data):
//   add, mul   every + - * between scalars (a fused multiply-add is one of each)
//   div        true divisions (none in the kernels' form; kept for other scalars)
//   minmax     min / max / clamp selections and the comparisons of the limit code
//   trans      sin, cos, exp2/exp, rsqrt, rcp (quarter-rate VALU instructions)
// flops = add + mul + div + minmax + trans.
//
// Only tests/, bench.py's reporting (it reads the committed
// profiles/flops_per_env_step.json, not this library) and tools may use this file;
// the product never does.  Build: make -C oracle libflopcount.so
#include <cstdint>
#include <cstring>
#include <string>

#include "../gym_roboy_amd/csrc/msj_build.hpp"

namespace {
struct Tally { uint64_t add = 0, mul = 0, div = 0, minmax = 0, trans = 0; };
Tally g_tally;
}  // namespace

struct Counted {
    double v;
    Counted() : v(0.0) {}
    Counted(double x) : v(x) {}
    Counted(float x) : v(x) {}
    Counted(int x) : v(x) {}
};
inline Counted operator+(Counted a, Counted b) { ++g_tally.add; return Counted(a.v + b.v); }
inline Counted operator-(Counted a, Counted b) { ++g_tally.add; return Counted(a.v - b.v); }
inline Counted operator*(Counted a, Counted b) { ++g_tally.mul; return Counted(a.v * b.v); }
inline Counted operator/(Counted a, Counted b) { ++g_tally.div; return Counted(a.v / b.v); }
inline Counted &operator+=(Counted &a, Counted b) { a = a + b; return a; }
inline Counted &operator-=(Counted &a, Counted b) { a = a - b; return a; }
inline Counted &operator*=(Counted &a, Counted b) { a = a * b; return a; }
inline Counted operator-(Counted a) { return Counted(-a.v); }             // a source modifier on the GPU: free
inline bool operator<(Counted a, Counted b) { ++g_tally.minmax; return a.v < b.v; }
inline bool operator>(Counted a, Counted b) { ++g_tally.minmax; return a.v > b.v; }

namespace rb {
template <> struct Fast<Counted> {
    static void sincos(Counted x, Counted &s, Counted &c) { g_tally.trans += 2; s = Counted(::sin(x.v)); c = Counted(::cos(x.v)); }
    static Counted exp(Counted x) { ++g_tally.trans; return Counted(::exp(x.v)); }
    static Counted exp2(Counted x) { ++g_tally.trans; return Counted(::exp2(x.v)); }
    static Counted rsqrt(Counted x) { ++g_tally.trans; return Counted(1.0 / ::sqrt(x.v)); }
    static Counted rcp(Counted x) { ++g_tally.trans; return Counted(1.0 / x.v); }
};
// min / max / clamp are single VALU selections (v_min / v_max / v_med3): one tally
// each instead of the comparison operators they are written with on the host
template <> inline Counted tmin<Counted>(Counted a, Counted b) { ++g_tally.minmax; return a.v < b.v ? a : b; }
template <> inline Counted tmax<Counted>(Counted a, Counted b) { ++g_tally.minmax; return a.v > b.v ? a : b; }
template <> inline Counted tclamp<Counted>(Counted x, Counted lo, Counted hi) {
    ++g_tally.minmax;
    return x.v < lo.v ? lo : (x.v > hi.v ? hi : x);
}
}  // namespace rb

// out[0..4] = add, mul, div, minmax, trans of ONE env step from (q, qd) with set-points sp
// (metres of tendon set-point offset, i.e. already scaled); q, qd are advanced in place.
extern "C" int fc_msj_step(const rb_robot_desc *d, double step_size, int nsub, int integ,
                           double *q, double *qd, const double *sp, uint64_t *out, unsigned char *feasible) {
    std::string err;
    if (d->n_t != 8) return RB_EUNSUPPORTED;
    rb::MsjConst<Counted, 8> c;
    const int rc = rb::msj_build<Counted, 8>(d, step_size, nsub, &c, err);
    if (rc) return rc;
    Counted qq[3], vv[3], ss[8];
    for (int j = 0; j < 3; ++j) { qq[j] = Counted(q[j]); vv[j] = Counted(qd[j]); }
    for (int k = 0; k < 8; ++k) ss[k] = Counted(sp[k]);
    g_tally = Tally();
    const bool ok = integ == 0 ? rb::MsjModel<Counted, 8>::template step<0>(c, qq, vv, ss)
                               : rb::MsjModel<Counted, 8>::template step<1>(c, qq, vv, ss);
    out[0] = g_tally.add; out[1] = g_tally.mul; out[2] = g_tally.div; out[3] = g_tally.minmax; out[4] = g_tally.trans;
    for (int j = 0; j < 3; ++j) { q[j] = qq[j].v; qd[j] = vv[j].v; }
    if (feasible) *feasible = ok ? 1 : 0;
    return RB_OK;
}

// ---------------------------------------------------------------------------------------------
// Joint-tree robots: scalar restatement of what csrc/tree_aba.hpp evaluates per env (articulated-
// body algorithm in world coordinates about the world origin, tendons reduced to their link
// crossings, the muscle model on the scaled strain), templated on the scalar so that it runs in
// fp64 (a third witness beside the numpy / C oracles, tests/test_flop_count.py) and with the
// tallying scalar (the flop count of configs[3]).  One lane's worth of arithmetic per quantity:
// the kernel's octets repeat some of it in several lanes, which is not algorithmic work.
#include <vector>
namespace {
template <typename T> struct Vec3 { T x, y, z; };
template <typename T> Vec3<T> operator+(Vec3<T> a, Vec3<T> b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
template <typename T> Vec3<T> operator-(Vec3<T> a, Vec3<T> b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
template <typename T> Vec3<T> operator*(Vec3<T> a, T s) { return {a.x * s, a.y * s, a.z * s}; }
template <typename T> T dot(Vec3<T> a, Vec3<T> b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
template <typename T> Vec3<T> cross(Vec3<T> a, Vec3<T> b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
template <typename T> struct Mat3 { T m[9]; };
template <typename T> Vec3<T> mul(const Mat3<T> &a, Vec3<T> v) {
    return {a.m[0] * v.x + a.m[1] * v.y + a.m[2] * v.z, a.m[3] * v.x + a.m[4] * v.y + a.m[5] * v.z, a.m[6] * v.x + a.m[7] * v.y + a.m[8] * v.z};
}
template <typename T> Vec3<T> mulT(const Mat3<T> &a, Vec3<T> v) {
    return {a.m[0] * v.x + a.m[3] * v.y + a.m[6] * v.z, a.m[1] * v.x + a.m[4] * v.y + a.m[7] * v.z, a.m[2] * v.x + a.m[5] * v.y + a.m[8] * v.z};
}
template <typename T> Mat3<T> mul(const Mat3<T> &a, const Mat3<T> &b) {
    Mat3<T> o;
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) o.m[3 * r + c] = a.m[3 * r] * b.m[c] + a.m[3 * r + 1] * b.m[3 + c] + a.m[3 * r + 2] * b.m[6 + c];
    return o;
}
template <typename T> Vec3<T> symmul(const T *s, Vec3<T> v) {
    return {s[0] * v.x + s[3] * v.y + s[4] * v.z, s[3] * v.x + s[1] * v.y + s[5] * v.z, s[4] * v.x + s[5] * v.y + s[2] * v.z};
}

template <typename T>
void tree_accel(const rb_robot_desc *d, const T *q, const T *qd, const T *sp, T *qdd) {
    using V = Vec3<T>;
    using F = rb::Fast<T>;
    const int nq = d->n_q, nt = d->n_t;
    const double log2e = 1.4426950408889634, sc = std::sqrt(log2e) / d->fl_width;
    std::vector<Mat3<T>> R(nq);
    std::vector<V> p(nq), z(nq), sl(nq), w(nq), vo(nq), ca(nq), cl(nq);
    const V zero = {T(0), T(0), T(0)};
    for (int i = 0; i < nq; ++i) {
        const int par = d->parent[i];
        Mat3<T> Rp = {{T(1), T(0), T(0), T(0), T(1), T(0), T(0), T(0), T(1)}};
        V pp = zero, wp = zero, vop = zero;
        if (par >= 0) { Rp = R[par]; pp = p[par]; wp = w[par]; vop = vo[par]; }
        const V ax = {T(d->axis[3 * i]), T(d->axis[3 * i + 1]), T(d->axis[3 * i + 2])};
        const V org = {T(d->origin[3 * i]), T(d->origin[3 * i + 1]), T(d->origin[3 * i + 2])};
        T sn, cs;
        F::sincos(q[i], sn, cs);
        const T oc = T(1) - cs;
        const Mat3<T> rot = {{T(1) - oc * (ax.y * ax.y + ax.z * ax.z), -sn * ax.z + oc * ax.x * ax.y, sn * ax.y + oc * ax.x * ax.z,
                              sn * ax.z + oc * ax.x * ax.y, T(1) - oc * (ax.x * ax.x + ax.z * ax.z), -sn * ax.x + oc * ax.y * ax.z,
                              -sn * ax.y + oc * ax.x * ax.z, sn * ax.x + oc * ax.y * ax.z, T(1) - oc * (ax.x * ax.x + ax.y * ax.y)}};
        if (par >= 0) { R[i] = mul(Rp, rot); p[i] = pp + mul(Rp, org); z[i] = mul(Rp, ax); }
        else { R[i] = rot; p[i] = org; z[i] = ax; }          // identity parent: nothing to multiply
        sl[i] = cross(p[i], z[i]);
        w[i] = wp + z[i] * qd[i];
        vo[i] = vop + sl[i] * qd[i];
        ca[i] = cross(wp, z[i]) * qd[i];
        cl[i] = (cross(w[i], sl[i]) + cross(vo[i], z[i])) * qd[i];
    }
    // tendons
    const T kps = T(d->kp / sc), pe_k2s = T(log2e * d->kpe / (d->e0 * sc)), inv_pe_den = T(1.0 / (std::exp(d->kpe) - 1.0));
    const double c2l_ = (1.0 + 1.0 / d->fv_a) / (d->fv_n - 1.0);
    const T fv_c2s = T(-1.0 / d->fv_a), fv_k = T(1.0 + 1.0 / d->fv_a), fv_c1l = T(d->fv_n * c2l_), fv_c2l = T(c2l_);
    std::vector<V> fa(nq, zero), fl_(nq, zero);     // tendon wrench sums per link, as they enter p^A
    std::vector<double> org(3 * nq);
    for (int i = 0; i < nq; ++i)
        for (int a = 0; a < 3; ++a) org[3 * i + a] = (d->parent[i] < 0 ? 0.0 : org[3 * d->parent[i] + a]) + d->origin[3 * i + a];
    auto point = [&](int link, int v, V &x, V &xd) {
        const V r = {T(d->vp_pos[3 * v]), T(d->vp_pos[3 * v + 1]), T(d->vp_pos[3 * v + 2])};
        if (link < 0) { x = r; xd = zero; return; }
        x = p[link] + mul(R[link], r);
        xd = vo[link] + cross(w[link], x);
    };
    for (int k = 0; k < nt; ++k) {
        const int v0 = d->vp_offset[k], v1 = d->vp_offset[k + 1];
        double l0 = 0.0, lconst = 0.0;
        for (int v = v0; v + 1 < v1; ++v) {          // host-side constants (fp64, not counted: tree_build does this once)
            const int la = d->vp_link[v], lb = d->vp_link[v + 1];
            double s = 0.0, s2 = 0.0;
            for (int a = 0; a < 3; ++a) {
                const double xa = (la < 0 ? 0.0 : org[3 * la + a]) + d->vp_pos[3 * v + a], xb = (lb < 0 ? 0.0 : org[3 * lb + a]) + d->vp_pos[3 * (v + 1) + a];
                s += (xb - xa) * (xb - xa);
                const double dl = d->vp_pos[3 * (v + 1) + a] - d->vp_pos[3 * v + a];
                s2 += dl * dl;
            }
            l0 += std::sqrt(s);
            if (la == lb) lconst += std::sqrt(s2);
        }
        T len = T(0), ldot = T(0);
        struct Cr { int la, lb; V n, u; };
        std::vector<Cr> crs;
        for (int v = v0; v + 1 < v1; ++v) {
            const int la = d->vp_link[v], lb = d->vp_link[v + 1];
            if (la == lb) continue;
            V xa, va, xb, vb;
            point(la, v, xa, va); point(lb, v + 1, xb, vb);
            const V dd = xb - xa;
            const T d2 = dot(dd, dd), inv = F::rsqrt(d2);
            const V u = dd * inv;
            len = len + d2 * inv;
            ldot = ldot + dot(u, vb - va);
            crs.push_back({la, lb, cross(xa, u), u});
        }
        const T es = len * T(sc / l0) + T(sc * (lconst / l0 - 1.0));
        const T act = rb::tclamp(kps * es - T(d->kp * d->setpoint_scale / l0) * sp[k], T(0), T(1));
        const T fl = F::exp2(-(es * es));
        const T v = ldot * T(1.0 / (d->v_max * l0));
        const T vp = rb::tmax(v, T(0)), pcl = rb::tclamp(v + T(1), T(0), T(1));
        const T num = fv_c1l * vp + pcl, den = fv_c2l * vp + (fv_c2s * pcl + fv_k);
        const T fpe = rb::tmax(F::exp2(pe_k2s * es) * inv_pe_den - inv_pe_den, T(0));
        const T Fk = T(d->f_max[k]) * ((act * fl) * num * F::rcp(den) + fpe);
        for (const Cr &c : crs) {
            const V Wn = c.n * Fk, Wf = c.u * Fk;
            if (c.la >= 0) { fa[c.la] = fa[c.la] - Wn; fl_[c.la] = fl_[c.la] - Wf; }
            if (c.lb >= 0) { fa[c.lb] = fa[c.lb] + Wn; fl_[c.lb] = fl_[c.lb] + Wf; }
        }
    }
    // backward pass: 6x6 symmetric articulated inertia as AA (sym 6), AL (3x3), LL (sym 6)
    struct Art { T AA[6], AL[9], LL[6]; V pa, pl; };
    std::vector<Art> A(nq);
    std::vector<V> Ua(nq), Ul(nq);
    std::vector<T> invD(nq), u(nq);
    for (int i = 0; i < nq; ++i) {
        Art &a = A[i];
        const T m = T(d->mass[i]);
        const double *I6d = d->inertia + 6 * i;
        if (d->mass[i] != 0.0 || I6d[0] != 0.0 || I6d[1] != 0.0 || I6d[2] != 0.0) {
            const T I6[6] = {T(I6d[0]), T(I6d[1]), T(I6d[2]), T(I6d[3]), T(I6d[4]), T(I6d[5])};
            const V com = {T(d->com[3 * i]), T(d->com[3 * i + 1]), T(d->com[3 * i + 2])};
            const V cw = p[i] + mul(R[i], com);
            Mat3<T> RI;
            for (int r = 0; r < 3; ++r) {
                const V row = {R[i].m[3 * r], R[i].m[3 * r + 1], R[i].m[3 * r + 2]};
                const V ri = symmul(I6, row);
                RI.m[3 * r] = ri.x; RI.m[3 * r + 1] = ri.y; RI.m[3 * r + 2] = ri.z;
            }
            auto rowdot = [&](int r1, int r2) { return RI.m[3 * r1] * R[i].m[3 * r2] + RI.m[3 * r1 + 1] * R[i].m[3 * r2 + 1] + RI.m[3 * r1 + 2] * R[i].m[3 * r2 + 2]; };
            const T c2 = dot(cw, cw);
            a.AA[0] = rowdot(0, 0) + m * (c2 - cw.x * cw.x); a.AA[1] = rowdot(1, 1) + m * (c2 - cw.y * cw.y); a.AA[2] = rowdot(2, 2) + m * (c2 - cw.z * cw.z);
            a.AA[3] = rowdot(0, 1) - m * cw.x * cw.y; a.AA[4] = rowdot(0, 2) - m * cw.x * cw.z; a.AA[5] = rowdot(1, 2) - m * cw.y * cw.z;
            const V h = cw * m;
            const T AL[9] = {T(0), -h.z, h.y, h.z, T(0), -h.x, -h.y, h.x, T(0)};
            for (int e = 0; e < 9; ++e) a.AL[e] = AL[e];
            a.LL[0] = m; a.LL[1] = m; a.LL[2] = m; a.LL[3] = T(0); a.LL[4] = T(0); a.LL[5] = T(0);
            const V Iva = symmul(a.AA, w[i]) + cross(h, vo[i]), Ivl = vo[i] * m - cross(h, w[i]);
            a.pa = cross(w[i], Iva) + cross(vo[i], Ivl) + fa[i];
            a.pl = cross(w[i], Ivl) + fl_[i];
        } else {
            for (int e = 0; e < 6; ++e) { a.AA[e] = T(0); a.LL[e] = T(0); }
            for (int e = 0; e < 9; ++e) a.AL[e] = T(0);
            a.pa = fa[i]; a.pl = fl_[i];
        }
    }
    for (int i = nq - 1; i >= 0; --i) {
        Art &a = A[i];
        Mat3<T> al;
        for (int e = 0; e < 9; ++e) al.m[e] = a.AL[e];
        Ua[i] = symmul(a.AA, z[i]) + mul(al, sl[i]);
        Ul[i] = mulT(al, z[i]) + symmul(a.LL, sl[i]);
        const T D = dot(z[i], Ua[i]) + dot(sl[i], Ul[i]) + T(d->armature[i]);
        invD[i] = F::rcp(D);
        u[i] = -T(d->damping[i]) * qd[i] - (dot(z[i], a.pa) + dot(sl[i], a.pl));
        const int par = d->parent[i];
        if (par < 0) continue;
        const V Ka = Ua[i] * invD[i], Kl = Ul[i] * invD[i];
        const V ua = Ua[i], ul = Ul[i];
        T AA[6] = {a.AA[0] - Ka.x * ua.x, a.AA[1] - Ka.y * ua.y, a.AA[2] - Ka.z * ua.z, a.AA[3] - Ka.x * ua.y, a.AA[4] - Ka.x * ua.z, a.AA[5] - Ka.y * ua.z};
        T AL[9] = {a.AL[0] - Ka.x * ul.x, a.AL[1] - Ka.x * ul.y, a.AL[2] - Ka.x * ul.z, a.AL[3] - Ka.y * ul.x, a.AL[4] - Ka.y * ul.y, a.AL[5] - Ka.y * ul.z,
                   a.AL[6] - Ka.z * ul.x, a.AL[7] - Ka.z * ul.y, a.AL[8] - Ka.z * ul.z};
        T LL[6] = {a.LL[0] - Kl.x * ul.x, a.LL[1] - Kl.y * ul.y, a.LL[2] - Kl.z * ul.z, a.LL[3] - Kl.x * ul.y, a.LL[4] - Kl.x * ul.z, a.LL[5] - Kl.y * ul.z};
        Mat3<T> ial;
        for (int e = 0; e < 9; ++e) ial.m[e] = AL[e];
        const T ud = u[i] * invD[i];
        const V na = a.pa + symmul(AA, ca[i]) + mul(ial, cl[i]) + ua * ud;
        const V nl = a.pl + mulT(ial, ca[i]) + symmul(LL, cl[i]) + ul * ud;
        Art &pr = A[par];
        for (int e = 0; e < 6; ++e) { pr.AA[e] = pr.AA[e] + AA[e]; pr.LL[e] = pr.LL[e] + LL[e]; }
        for (int e = 0; e < 9; ++e) pr.AL[e] = pr.AL[e] + AL[e];
        pr.pa = pr.pa + na; pr.pl = pr.pl + nl;
    }
    std::vector<V> aa(nq), al(nq);
    for (int i = 0; i < nq; ++i) {
        const int par = d->parent[i];
        V pa_ = zero, pl_ = {T(-d->gravity[0]), T(-d->gravity[1]), T(-d->gravity[2])};
        if (par >= 0) { pa_ = aa[par]; pl_ = al[par]; }
        pa_ = pa_ + ca[i]; pl_ = pl_ + cl[i];
        qdd[i] = (u[i] - (dot(Ua[i], pa_) + dot(Ul[i], pl_))) * invD[i];
        aa[i] = pa_ + z[i] * qdd[i]; al[i] = pl_ + sl[i] * qdd[i];
    }
}

template <typename T>
bool tree_step(const rb_robot_desc *d, double step_size, int nsub, int integ, T *q, T *qd, const T *sp) {
    const int nq = d->n_q;
    const T h = T(step_size / nsub);
    bool feasible = true;
    std::vector<T> a(nq), k1q(nq), k1v(nq), k2q(nq), k2v(nq), k3q(nq), k3v(nq), k4q(nq), k4v(nq), qs(nq);
    auto sat = [&](T v, int j) { return rb::tclamp(v, T(-d->qd_max[j]), T(d->qd_max[j])); };
    for (int sub = 0; sub < nsub; ++sub) {
        if (integ == 0) {
            tree_accel<T>(d, q, qd, sp, a.data());
            for (int j = 0; j < nq; ++j) { qd[j] = sat(qd[j] + h * a[j], j); q[j] = q[j] + h * qd[j]; }
        } else {
            const T hh = T(0.5) * h, h6 = h * T(1.0 / 6.0);
            for (int j = 0; j < nq; ++j) k1q[j] = sat(qd[j], j);
            tree_accel<T>(d, q, k1q.data(), sp, k1v.data());
            for (int j = 0; j < nq; ++j) { k2q[j] = sat(qd[j] + hh * k1v[j], j); qs[j] = q[j] + hh * k1q[j]; }
            tree_accel<T>(d, qs.data(), k2q.data(), sp, k2v.data());
            for (int j = 0; j < nq; ++j) { k3q[j] = sat(qd[j] + hh * k2v[j], j); qs[j] = q[j] + hh * k2q[j]; }
            tree_accel<T>(d, qs.data(), k3q.data(), sp, k3v.data());
            for (int j = 0; j < nq; ++j) { k4q[j] = sat(qd[j] + h * k3v[j], j); qs[j] = q[j] + h * k3q[j]; }
            tree_accel<T>(d, qs.data(), k4q.data(), sp, k4v.data());
            for (int j = 0; j < nq; ++j) {
                q[j] = q[j] + h6 * (k1q[j] + T(2) * k2q[j] + T(2) * k3q[j] + k4q[j]);
                qd[j] = qd[j] + h6 * (k1v[j] + T(2) * k2v[j] + T(2) * k3v[j] + k4v[j]);
            }
        }
        for (int j = 0; j < nq; ++j) {
            T v = sat(qd[j], j);
            const bool over = q[j] > T(d->q_hi[j]), under = q[j] < T(d->q_lo[j]);
            if (over) { q[j] = T(d->q_hi[j]); v = rb::tmin(v, T(0)); }
            if (under) { q[j] = T(d->q_lo[j]); v = rb::tmax(v, T(0)); }
            qd[j] = v;
            feasible = feasible && !(over || under);
        }
    }
    return feasible;
}
}  // namespace

// same contract as fc_msj_step, for any joint tree: out[0..4] = add, mul, div, minmax, trans of one env step
extern "C" int fc_tree_step(const rb_robot_desc *d, double step_size, int nsub, int integ,
                            double *q, double *qd, const double *sp, uint64_t *out, unsigned char *feasible) {
    std::vector<Counted> qq(d->n_q), vv(d->n_q), ss(d->n_t);
    for (int j = 0; j < d->n_q; ++j) { qq[j] = Counted(q[j]); vv[j] = Counted(qd[j]); }
    for (int k = 0; k < d->n_t; ++k) ss[k] = Counted(sp[k]);
    g_tally = Tally();
    const bool ok = tree_step<Counted>(d, step_size, nsub, integ, qq.data(), vv.data(), ss.data());
    out[0] = g_tally.add; out[1] = g_tally.mul; out[2] = g_tally.div; out[3] = g_tally.minmax; out[4] = g_tally.trans;
    for (int j = 0; j < d->n_q; ++j) { q[j] = qq[j].v; qd[j] = vv[j].v; }
    if (feasible) *feasible = ok ? 1 : 0;
    return RB_OK;
}
