// msj_math.hpp - per-env arithmetic of the ball-joint ("MSJ class") tendon robot.
//
// Model: DESIGN.md §2 (spec) and §4 (this closed form).  A robot of this class
// is one rigid body on a ball joint written as three chained revolutes x-y-z
// through the base origin, pulled by NT tendons that each have exactly one
// moving segment (last base via-point A_k -> first body via-point B_k); every
// other segment has constant length and is folded into `lc`.  Everything is
// evaluated in the BODY frame, where the inertia about the joint centre and
// the body-side via-points are constants:
//
//   R    = Rx(q0) Ry(q1) Rz(q2)                    body -> world
//   zeta = R^T [z0 z1 z2] = [(c1c2,-c1s2,s1), (s2,c2,0), (0,0,1)]
//   w_b  = zeta qd                                  angular velocity
//   a_k  = R^T A_k,  d = B_k - a_k,  u = d/|d|,  l_k = |d| + lc_k
//   w_k  = B_k x u;  dl_k/dt = w_b . w_k;  row k of the cable-length Jacobian
//          is L_kj = zeta_j . w_k (never formed: -L^T F = zeta^T (-sum F_k w_k))
//   M    = zeta^T I_O zeta + diag(armature)
//   bias = zeta^T ( I_O (zeta_dot qd) + w_b x I_O w_b - (m c) x R^T g )
//   qdd  = M^-1 ( zeta^T(-sum F_k w_k) - D qd - bias )
//
// The reference has no counterpart (its physics is the external CARDSflow
// step behind ros_simulation_client.py:48-60); the CPU oracle
// (oracle/physics_np.py, oracle/roboy_oracle.c) evaluates the same model with
// the generic tree algorithms and is what this file is checked against.
//
// The header is plain C++ templated on the scalar type so the same source is
// compiled by hipcc for the kernels and by g++ in tests/ (host double/float)
// to check the derivation without a GPU.
#pragma once
#include "rtc_compat.hpp"

#if defined(__HIPCC__)
#define RB_HD __host__ __device__ __forceinline__
#else
#define RB_HD inline
#endif


namespace rb {

// ---- scalar helpers: fast hardware forms on the device, libm on the host ----
template <typename T> struct Fast;
template <> struct Fast<double> {
    static RB_HD void sincos(double x, double &s, double &c) { s = ::sin(x); c = ::cos(x); }
    static RB_HD double exp(double x) { return ::exp(x); }
    static RB_HD double exp2(double x) { return ::exp2(x); }
    static RB_HD double rsqrt(double x) { return 1.0 / ::sqrt(x); }
    static RB_HD double rcp(double x) { return 1.0 / x; }
};
template <> struct Fast<float> {
    static RB_HD void sincos(float x, float &s, float &c) {
#if defined(__HIP_DEVICE_COMPILE__)
        s = __sinf(x); c = __cosf(x);          // v_sin_f32 / v_cos_f32
#else
        s = ::sinf(x); c = ::cosf(x);
#endif
    }
    static RB_HD float exp(float x) {
#if defined(__HIP_DEVICE_COMPILE__)
        return __expf(x);                       // v_exp_f32
#else
        return ::expf(x);
#endif
    }
    static RB_HD float exp2(float x) {
#if defined(__HIP_DEVICE_COMPILE__)
        return __builtin_amdgcn_exp2f(x);       // bare v_exp_f32
#else
        return ::exp2f(x);
#endif
    }
    static RB_HD float rsqrt(float x) {
#if defined(__HIP_DEVICE_COMPILE__)
        return __builtin_amdgcn_rsqf(x);        // v_rsq_f32 (1 ulp)
#else
        return 1.0f / ::sqrtf(x);
#endif
    }
    static RB_HD float rcp(float x) {
#if defined(__HIP_DEVICE_COMPILE__)
        return __builtin_amdgcn_rcpf(x);        // v_rcp_f32 (1 ulp)
#else
        return 1.0f / x;
#endif
    }
};

// min / max / clamp: on the device, for float, one VALU instruction each (v_min_f32 /
// v_max_f32 / v_med3_f32) instead of a compare-select pair
template <typename T> RB_HD T tmin(T a, T b) {
#if defined(__HIP_DEVICE_COMPILE__)
    if constexpr (std::is_same<T, float>::value) return __builtin_fminf(a, b);
#endif
    return a < b ? a : b;
}
template <typename T> RB_HD T tmax(T a, T b) {
#if defined(__HIP_DEVICE_COMPILE__)
    if constexpr (std::is_same<T, float>::value) return __builtin_fmaxf(a, b);
#endif
    return a > b ? a : b;
}
template <typename T> RB_HD T tclamp(T x, T lo, T hi) {
#if defined(__HIP_DEVICE_COMPILE__)
    if constexpr (std::is_same<T, float>::value) return __builtin_amdgcn_fmed3f(x, lo, hi);
#endif
    return tmin(tmax(x, lo), hi);
}

// ---- per-robot constants (wave-uniform: fetched by scalar loads into SGPRs) ----
// One 16-scalar record per tendon, so a rolled tendon loop needs a single
// s_load_dwordx16 per trip and only ~16 SGPRs of tendon constants are live.
// Lengths enter the muscle model through the scaled strain  es = s (l/l0 - 1)  with
// s = sqrt(log2 e)/fl_width, so that f_L = exp2(-es^2) needs no further multiply; the other
// users of the strain carry 1/s in their own constants (kps, pe_k2s).
template <typename T>
struct alignas(64) MsjTendon {
    T A[3];        // last base via-point, world = base frame
    T Bv[3];       // first body via-point B (body frame) / (v_max * rest length): a x Bv is the torque arm
                   // times |d| in units that make (w . (a x Bv)) / |d| the normalised lengthening rate v
    T B2[3];       // -2 B  (|B - a|^2 = ab2 + a . B2)
    T ab2;         // |A|^2 + |B|^2
    T il0s;        // s / rest length
    T elcs;        // s (lc / l0 - 1)  (lc = summed length of the segments that do not move)
    T ksg;         // kp * setpoint_scale / rest length: set-point -> activation offset u = ksg * setpoint
    T fmaxv;       // maximum isometric force * (v_max * rest length)  (undoes the scale of Bv in the torque)
    T pad[2];
};

template <typename T, int NT>
struct MsjConst {
    MsjTendon<T> ten[NT];
    T IO[6];          // inertia about the joint centre, body frame: xx,yy,zz,xy,xz,yz
    T mc[3];          // mass * centre of mass (body frame)
    T g[3];           // gravity, world
    T arm[3], damp[3], qlo[3], qhi[3], qdmax[3];
    T kps;            // kp / s: activation per unit of scaled strain
    T pe_k2s;         // log2(e) * kpe / (e0 s)          (f_PE = (exp2(pe_k2s * es) - 1) * inv_pe_den)
    T inv_pe_den;     // 1 / (exp(kpe) - 1)
    T fv_c1l, fv_c2l; // lengthening branch of f_V; the shortening branch is (1 + v)/(1 + fv_c2s v)
    T fv_c2s;
    T fv_k;           // 1 - fv_c2s
    T h;              // integrator substep
    int32_t nsub;
    int32_t simple;   // 1: principal-axis inertia, COM on body z, gravity along world z (fast path)
    int32_t nt;       // tendons in use (<= NT); only the run-time-count kernels (UNROLL = 0) read it
};

// set-points held in a per-lane register array (compile-time indices only)
template <typename T, int NT>
struct SpArray {
    const T *v;
    RB_HD T operator()(int k) const { return v[k]; }
};

template <typename T, int NT>
struct MsjModel {
    using C = MsjConst<T, NT>;

    // body frame of one env: rotation rows, joint-axis sines/cosines, angular velocity
    struct Frame {
        T s0, c0, s1, c1, s2, c2;
        T r00, r01, r02, r10, r11, r12, r20, r21, r22;
        T wx, wy, wz;
    };

    static RB_HD Frame frame(const T q[3], const T qd[3]) {
        Frame f;
        Fast<T>::sincos(q[0], f.s0, f.c0);
        Fast<T>::sincos(q[1], f.s1, f.c1);
        Fast<T>::sincos(q[2], f.s2, f.c2);
        // R = Rx Ry Rz, rows
        f.r00 = f.c1 * f.c2; f.r01 = -f.c1 * f.s2; f.r02 = f.s1;
        f.r10 = f.c0 * f.s2 + f.s0 * f.s1 * f.c2; f.r11 = f.c0 * f.c2 - f.s0 * f.s1 * f.s2; f.r12 = -f.s0 * f.c1;
        f.r20 = f.s0 * f.s2 - f.c0 * f.s1 * f.c2; f.r21 = f.s0 * f.c2 + f.c0 * f.s1 * f.s2; f.r22 = f.c0 * f.c1;
        // joint axes in the body frame: zeta0 = row 0 of R, zeta1 = (s2,c2,0), zeta2 = e_z
        f.wx = f.r00 * qd[0] + f.s2 * qd[1];
        f.wy = f.r01 * qd[0] + f.c2 * qd[1];
        f.wz = f.r02 * qd[0] + qd[2];
        return f;
    }

    // one tendon: routing, Hill-type force, torque about the joint centre (body
    // frame) subtracted from (tx,ty,tz).  `u` is the tendon's activation offset
    // ksg * set-point (prescale()).
    static RB_HD void tendon(const C &c, const Frame &f, const MsjTendon<T> &t, T u, T &tx, T &ty, T &tz) {
        // a = R^T A (body frame).  With |a| = |A|:  |B - a|^2 = (|A|^2 + |B|^2) - 2 a.B,
        // and the torque arm w = B x (B - a)/|d| = (a x B)/|d|, so neither the
        // difference vector nor the unit vector is formed.
        const T ax = f.r00 * t.A[0] + f.r10 * t.A[1] + f.r20 * t.A[2];
        const T ay = f.r01 * t.A[0] + f.r11 * t.A[1] + f.r21 * t.A[2];
        const T az = f.r02 * t.A[0] + f.r12 * t.A[1] + f.r22 * t.A[2];
        const T d2 = ax * t.B2[0] + (ay * t.B2[1] + (az * t.B2[2] + t.ab2));
        // m = (a x B)/(v_max l0); the torque arm is m/|d| up to that scale, which fmaxv undoes
        const T mx = ay * t.Bv[2] - az * t.Bv[1];
        const T my = az * t.Bv[0] - ax * t.Bv[2];
        const T mz = ax * t.Bv[1] - ay * t.Bv[0];
        tendon_force(c, f, t, u, d2, mx, my, mz, tx, ty, tz);
    }

    // the tendon behind its routing: from d2 = |B - a|^2 and m = (a x B)/(v_max l0) to the Hill-type force and
    // the torque about the joint centre
    static RB_HD void tendon_force(const C &c, const Frame &f, const MsjTendon<T> &t, T u, T d2, T mx, T my, T mz,
                                   T &tx, T &ty, T &tz) {
        const T inv = Fast<T>::rsqrt(d2);
        const T v = (f.wx * mx + f.wy * my + f.wz * mz) * inv;      // (dl/dt) / (v_max l0)
        // Hill-type muscle on the scaled strain es = s (l/l0 - 1) = s |d|/l0 + s (lc/l0 - 1):
        // activation = clamp(kp (e - (sigma/l0) setpoint), 0, 1) = clamp((kp/s) es - u, 0, 1)
        const T es = (d2 * inv) * t.il0s + t.elcs;
        const T act = tclamp(c.kps * es - u, T(0), T(1));
        const T fl = Fast<T>::exp2(-(es * es));
        // f_V = (1 + c1 v)/(1 + c2 v) per branch, written branch-free with v+ = max(v,0) and
        // p = 1 + clamp(v,-1,0) = clamp(1 + v, 0, 1)  (v+ = 0 or p = 1)
        const T vp = tmax(v, T(0)), p = tclamp(v + T(1), T(0), T(1));
        const T num = c.fv_c1l * vp + p;
        const T den = c.fv_c2l * vp + (c.fv_c2s * p + c.fv_k);
        const T rden = Fast<T>::rcp(den);              // den >= 1
        const T fpe = tmax(Fast<T>::exp2(c.pe_k2s * es) * c.inv_pe_den - c.inv_pe_den, T(0));
        const T Fs = (t.fmaxv * inv) * ((act * fl) * num * rden + fpe);     // tension (v_max l0) / |d|
        tx -= Fs * mx; ty -= Fs * my; tz -= Fs * mz;
    }

    // set-point (tendon length offset, the action box +-0.3 of msj_robot.py:15-16) -> activation offset
    static RB_HD T prescale(const C &c, int k, T setpoint) { return c.ten[k].ksg * setpoint; }

    // rigid body about the joint centre: qdd from the summed tendon torque.
    // c.simple (wave-uniform) marks the common case - principal-axis inertia,
    // centre of mass on the body z axis, gravity along world z - in which a
    // third of the terms vanish; the branch is a scalar one.
    static RB_HD void rigid_body(const C &c, const Frame &f, const T qd[3], T tx, T ty, T tz, T qdd[3]) {
        const T r00 = f.r00, r01 = f.r01, r02 = f.r02, s2 = f.s2, c2 = f.c2;
        const T Ixx = c.IO[0], Iyy = c.IO[1], Izz = c.IO[2];
        // zeta_dot qd
        const T z0x = -f.s1 * c2 * qd[1] - f.c1 * s2 * qd[2];
        const T z0y = f.s1 * s2 * qd[1] - f.c1 * c2 * qd[2];
        const T z0z = f.c1 * qd[1];
        const T bx = z0x * qd[0] + c2 * qd[2] * qd[1];
        const T by = z0y * qd[0] - s2 * qd[2] * qd[1];
        const T bz = z0z * qd[0];
        T m00, m01, m02, m11, m12, m22, vx, vy, vz;
        if (c.simple) {
            // I_O = diag(Ixx, Iyy, Izz), m c = (0, 0, mcz), g = (0, 0, gz)
            const T a0x = Ixx * r00, a0y = Iyy * r01, a0z = Izz * r02;
            const T a1x = Ixx * s2, a1y = Iyy * c2;
            m00 = r00 * a0x + r01 * a0y + r02 * a0z + c.arm[0];
            m01 = s2 * a0x + c2 * a0y;
            m02 = a0z;
            m11 = s2 * a1x + c2 * a1y + c.arm[1];
            m12 = T(0);
            m22 = Izz + c.arm[2];
            const T hx = Ixx * f.wx, hy = Iyy * f.wy, hz = Izz * f.wz;      // I_O w
            const T nx = Ixx * bx + (f.wy * hz - f.wz * hy);
            const T ny = Iyy * by + (f.wz * hx - f.wx * hz);
            const T nz = Izz * bz + (f.wx * hy - f.wy * hx);
            const T k = c.mc[2] * c.g[2];                                   // (m c) x R^T g
            vx = tx - k * f.r21 - nx;
            vy = ty + k * f.r20 - ny;
            vz = tz - nz;
        } else {
            const T Ixy = c.IO[3], Ixz = c.IO[4], Iyz = c.IO[5];
            // columns I_O zeta_j
            const T a0x = Ixx * r00 + Ixy * r01 + Ixz * r02;
            const T a0y = Ixy * r00 + Iyy * r01 + Iyz * r02;
            const T a0z = Ixz * r00 + Iyz * r01 + Izz * r02;
            const T a1x = Ixx * s2 + Ixy * c2;
            const T a1y = Ixy * s2 + Iyy * c2;
            const T a1z = Ixz * s2 + Iyz * c2;
            // M = zeta^T I_O zeta + armature
            m00 = r00 * a0x + r01 * a0y + r02 * a0z + c.arm[0];
            m01 = s2 * a0x + c2 * a0y;
            m02 = a0z;
            m11 = s2 * a1x + c2 * a1y + c.arm[1];
            m12 = a1z;
            m22 = Izz + c.arm[2];
            // angular momentum and gyroscopic term
            const T hx = a0x * qd[0] + a1x * qd[1] + Ixz * qd[2];
            const T hy = a0y * qd[0] + a1y * qd[1] + Iyz * qd[2];
            const T hz = a0z * qd[0] + a1z * qd[1] + Izz * qd[2];
            const T nx = Ixx * bx + Ixy * by + Ixz * bz + (f.wy * hz - f.wz * hy);
            const T ny = Ixy * bx + Iyy * by + Iyz * bz + (f.wz * hx - f.wx * hz);
            const T nz = Ixz * bx + Iyz * by + Izz * bz + (f.wx * hy - f.wy * hx);
            // gravity torque (m c) x R^T g
            const T gx = r00 * c.g[0] + f.r10 * c.g[1] + f.r20 * c.g[2];
            const T gy = r01 * c.g[0] + f.r11 * c.g[1] + f.r21 * c.g[2];
            const T gz = r02 * c.g[0] + f.r12 * c.g[1] + f.r22 * c.g[2];
            vx = tx + (c.mc[1] * gz - c.mc[2] * gy) - nx;
            vy = ty + (c.mc[2] * gx - c.mc[0] * gz) - ny;
            vz = tz + (c.mc[0] * gy - c.mc[1] * gx) - nz;
        }
        const T t0 = r00 * vx + r01 * vy + r02 * vz - c.damp[0] * qd[0];
        const T t1 = s2 * vx + c2 * vy - c.damp[1] * qd[1];
        const T t2 = vz - c.damp[2] * qd[2];
        // 3x3 SPD solve by the adjugate (one reciprocal)
        const T k00 = m11 * m22 - m12 * m12;
        const T k01 = m02 * m12 - m01 * m22;
        const T k02 = m01 * m12 - m02 * m11;
        const T k11 = m00 * m22 - m02 * m02;
        const T k12 = m01 * m02 - m00 * m12;
        const T k22 = m00 * m11 - m01 * m01;
        const T idet = Fast<T>::rcp(m00 * k00 + m01 * k01 + m02 * k02);
        qdd[0] = (k00 * t0 + k01 * t1 + k02 * t2) * idet;
        qdd[1] = (k01 * t0 + k11 * t1 + k12 * t2) * idet;
        qdd[2] = (k02 * t0 + k12 * t1 + k22 * t2) * idet;
    }

    // Acceleration with all NT tendons evaluated by this lane.  UNROLL = unroll
    // factor of the tendon loop (NT: straight-line code, most ILP; 1: rolled,
    // one scalar load of the tendon record per trip, fewest registers; 0: rolled
    // with the trip count c.nt <= NT read at run time, for robots of the class
    // with another tendon count than the instantiated one).
    // SP: set-point source, sp(k) (register array or LDS column).
    template <int UNROLL, typename SP>
    struct AccelAllTendons {
        const C &c;
        const SP &sp;
        RB_HD void operator()(const T q[3], const T qd[3], T qdd[3]) const {
            const Frame f = frame(q, qd);
            T tx = T(0), ty = T(0), tz = T(0);
            if (UNROLL == 0) {
#pragma unroll 1
                for (int k = 0; k < c.nt; ++k) tendon(c, f, c.ten[k], sp(k), tx, ty, tz);
            } else {
                constexpr int U = UNROLL > 0 ? UNROLL : 1;
#pragma unroll U
                for (int k = 0; k < NT; ++k) tendon(c, f, c.ten[k], sp(k), tx, ty, tz);
            }
            rigid_body(c, f, qd, tx, ty, tz, qdd);
        }
    };

    // ---- mirror pairs (two lanes per env; msj_kernels.hpp) --------------------------------------------------------------
    // A robot with a mirror plane - tendons in mirror-image pairs, symmetric body - seen in the mirror is the same robot
    // in the state sigma * q (the two joints whose axes lie in the mirror plane change sign), and the images of tendons
    // half[0..3] are its tendons half[0..3].  MIRROR = 0: the x-z plane, S = diag(1,-1,1), sigma = (-1,+1,-1);
    // MIRROR = 1: the y-z plane, S = diag(-1,1,1), sigma = (+1,-1,-1).  A torque is a pseudovector: through the mirror it
    // becomes det(S) S t = sigma * t (component-wise), which is how the two half sums are combined.
    template <int MIRROR> static RB_HD T mirror_sign(int j) { return (j == 2 || j == MIRROR) ? T(-1) : T(1); }
    // torque of the FIRST FOUR tendons of c (the host orders the table so that they are one half of the mirror pairs)
    static RB_HD void half_torque(const C &c, const Frame &f, const T u[4], T &tx, T &ty, T &tz) {
        tx = T(0); ty = T(0); tz = T(0);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            tendon(c, f, c.ten[k], u[k], tx, ty, tz);
#if defined(__HIP_DEVICE_COMPILE__)
            __builtin_amdgcn_sched_barrier(0);     // one tendon's temporaries at a time: 8 waves per SIMD are the latency cover, not ILP
#endif
        }
    }
    // own half + the partner's half seen through the mirror
    template <int MIRROR>
    static RB_HD void mirror_combine(T &tx, T &ty, T &tz, T px, T py, T pz) {
        tx = MIRROR == 0 ? tx - px : tx + px;
        ty = MIRROR == 0 ? ty + py : ty - py;
        tz = tz - pz;
    }

    static RB_HD void sat(const C &c, const T v[3], T out[3]) {
#pragma unroll
        for (int j = 0; j < 3; ++j) out[j] = tclamp(v[j], -c.qdmax[j], c.qdmax[j]);
    }

    // velocity saturation + joint limits; returns false when a limit was hit.  A joint beyond a limit is
    // clamped onto it and keeps only the inward part of its (saturated) velocity: the velocity box shrinks
    // to [-vmax, 0] at the upper limit and to [0, vmax] at the lower one, so one clamp does both.
    static RB_HD bool limit(const C &c, T q[3], T qd[3]) {
        bool ok = true;
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const bool over = q[j] > c.qhi[j], under = q[j] < c.qlo[j];
            qd[j] = tclamp(qd[j], under ? T(0) : -c.qdmax[j], over ? T(0) : c.qdmax[j]);
            q[j] = tclamp(q[j], c.qlo[j], c.qhi[j]);
            ok = ok && !(over || under);
        }
        return ok;
    }

    // one env step = nsub integrator substeps with the set-points held;
    // `accel(q, qd, qdd)` is one of the acceleration functors
    template <int INTEG, typename ACCEL>
    static RB_HD bool integrate(const C &c, T q[3], T qd[3], const ACCEL &accel) {
        bool feasible = true;
        const T h = c.h;
        for (int sub = 0; sub < c.nsub; ++sub) {
            if (INTEG == 0) {          // semi-implicit Euler
                T a[3];
                accel(q, qd, a);
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    qd[j] = tclamp(qd[j] + h * a[j], -c.qdmax[j], c.qdmax[j]);
                    q[j] = q[j] + h * qd[j];
                }
            } else {                   // RK4, stage velocities saturated
                T k1q[3], k1v[3], k2q[3], k2v[3], k3q[3], k3v[3], k4q[3], k4v[3], qs[3], vs[3];
                const T hh = T(0.5) * h;
                sat(c, qd, k1q);
                accel(q, k1q, k1v);
#pragma unroll
                for (int j = 0; j < 3; ++j) { qs[j] = q[j] + hh * k1q[j]; vs[j] = qd[j] + hh * k1v[j]; }
                sat(c, vs, k2q);
                accel(qs, k2q, k2v);
#pragma unroll
                for (int j = 0; j < 3; ++j) { qs[j] = q[j] + hh * k2q[j]; vs[j] = qd[j] + hh * k2v[j]; }
                sat(c, vs, k3q);
                accel(qs, k3q, k3v);
#pragma unroll
                for (int j = 0; j < 3; ++j) { qs[j] = q[j] + h * k3q[j]; vs[j] = qd[j] + h * k3v[j]; }
                sat(c, vs, k4q);
                accel(qs, k4q, k4v);
                const T h6 = h * T(1.0 / 6.0);
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    q[j] = q[j] + h6 * (k1q[j] + T(2) * k2q[j] + T(2) * k3q[j] + k4q[j]);
                    qd[j] = qd[j] + h6 * (k1v[j] + T(2) * k2v[j] + T(2) * k3v[j] + k4v[j]);
                }
            }
            feasible = limit(c, q, qd) && feasible;
        }
        return feasible;
    }

    // The same step with RK4's four stages as a ROLLED loop over running sums: 18 values live across an acceleration
    // (q, qd, the two weighted sums, the stage state) whatever the stage, against the 24-30 the unrolled integrate()
    // accumulates by its fourth stage, and a quarter of the code - for kernel forms that need 8 waves per SIMD
    // (<= 64 registers).  Another summation order than integrate(): equal to ~1 ulp, not bit for bit.
    template <int INTEG, typename ACCEL>
    static RB_HD bool integrate_acc(const C &c, T q[3], T qd[3], const ACCEL &accel) {
        if (INTEG == 0) return integrate<0>(c, q, qd, accel);
        bool feasible = true;
        const T h = c.h, h6 = c.h * T(1.0 / 6.0);
        for (int sub = 0; sub < c.nsub; ++sub) {
            T aq[3] = {T(0), T(0), T(0)}, av[3] = {T(0), T(0), T(0)}, qs[3], vs[3];
#pragma unroll
            for (int j = 0; j < 3; ++j) { qs[j] = q[j]; vs[j] = qd[j]; }
#pragma unroll 1
            for (int stage = 0; stage < 4; ++stage) {
                T kq[3], kv[3];
                sat(c, vs, kq);
                accel(qs, kq, kv);
                const T w = (stage == 0 || stage == 3) ? T(1) : T(2);        // weight of this stage in the sums
                const T a = stage < 2 ? T(0.5) * h : h;                       // step to the next stage's state
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    aq[j] += w * kq[j]; av[j] += w * kv[j];
                    qs[j] = q[j] + a * kq[j]; vs[j] = qd[j] + a * kv[j];
                }
            }
#pragma unroll
            for (int j = 0; j < 3; ++j) { q[j] = q[j] + h6 * aq[j]; qd[j] = qd[j] + h6 * av[j]; }
            feasible = limit(c, q, qd) && feasible;
        }
        return feasible;
    }

    // All NT tendons written out one after the other, each behind a scheduling barrier (device): one tendon's temporaries
    // at a time.  The acceleration of the "rolled stages" form (RS) of the kernels, below.
    struct AccelPinned {
        const C &c;
        const T *u;          // activation offsets (prescale() of the set-points)
        RB_HD void operator()(const T q[3], const T qd[3], T qdd[3]) const {
            const Frame f = frame(q, qd);
            T tx = T(0), ty = T(0), tz = T(0);
#pragma unroll
            for (int k = 0; k < NT; ++k) {
                tendon(c, f, c.ten[k], u[k], tx, ty, tz);
#if defined(__HIP_DEVICE_COMPILE__)
                __builtin_amdgcn_sched_barrier(0);
#endif
            }
            rigid_body(c, f, qd, tx, ty, tz, qdd);
        }
    };
    // RS: the integrator's stages as a rolled loop over running sums (integrate_acc), the tendons written out inside it
    // (AccelPinned).  UNROLL = RS in the kernels' template arguments selects it.
    static constexpr int RS = 9;
    template <int INTEG>
    static RB_HD bool step_rs(const C &c, T q[3], T qd[3], const T u[NT]) {
        return integrate_acc<INTEG>(c, q, qd, AccelPinned{c, u});
    }

    // sp(k): activation offset of tendon k (prescale() of its set-point)
    template <int INTEG, int UNROLL, typename SP>
    static RB_HD bool step_sp(const C &c, T q[3], T qd[3], const SP &sp) {
        return integrate<INTEG>(c, q, qd, AccelAllTendons<UNROLL, SP>{c, sp});
    }

    // sp: raw set-points; step_sp's SP source yields activation offsets (prescale())
    template <int INTEG, int UNROLL = NT>
    static RB_HD bool step(const C &c, T q[3], T qd[3], const T sp[NT]) {
        T u[NT];
#pragma unroll
        for (int k = 0; k < NT; ++k) u[k] = prescale(c, k, sp[k]);
        const SpArray<T, NT> src{u};
        return step_sp<INTEG, UNROLL>(c, q, qd, src);
    }
};

}  // namespace rb
