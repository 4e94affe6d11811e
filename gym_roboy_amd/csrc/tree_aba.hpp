// tree_aba.hpp - generic joint-tree robots (roboy-tendon-robot/1): E envs per WAVE,
// articulated-body algorithm in world coordinates, eight lanes per link.
//
// For robots outside the ball-joint class (msj_math.hpp) - e.g. the 20-DOF / 38-tendon upper
// body of BASELINE.json configs[3] - one lane cannot hold an env.  Round 1 gave each env a whole
// wavefront and built the joint-space mass matrix (CRBA + in-LDS Cholesky); its tree passes ran
// at 1-3 active lanes and the kernel was bound by wave-instruction issue (109 us at 8 192 envs,
// profiles/r1_b).  This file replaces it:
//
//   * forward dynamics by the articulated-body algorithm (Featherstone), O(n_q), with every
//     spatial quantity expressed in WORLD coordinates about the WORLD ORIGIN.  Then the
//     accumulation of articulated inertias / bias forces into the parent is a plain sum (no
//     coordinate transforms), and there is no mass matrix, no factorisation, no substitution;
//   * E (template, 2) envs share a wave, and in the level passes every link of a tree level gets
//     an OCTET of lanes: lanes 0-2 own the angular rows / components, lanes 4-6 the linear ones
//     (3 and 7 idle along).  Inside an octet values move by DPP only (quad rotations for cross
//     products, a half swap, 8-lane sums); vectors are kept in "rotated" component order
//     (k, k+1, k+2 for the lane that owns component k), which is what the quad rotations deliver;
//   * a chain of links keeps its octet from level to level (tree_build assigns the slots), so
//     parent -> child data (frame, velocity, acceleration) and child -> parent data (articulated
//     inertia row, bias force) travel in REGISTERS; LDS is touched on the dependent path only at
//     branch points.  The link's own spatial inertia is evaluated by its octet inside the
//     backward pass, so no 6x6 per link is ever stored;
//   * a tendon acts on the links only where it crosses from one link to another: segments
//     between via-points of the same link have constant length and their forces cancel, so
//     tree_build() folds them into a constant and keeps the "crossings" (1 per tendon on the
//     upper body), each of which exerts +-W = +-F (x_a x u ; u) on its two links.
//
// Phases of one acceleration evaluation:
//   P1  level by level from the root, octets:  R, p, joint axis s = (z ; p x z), spatial velocity
//       v = (w ; vO), velocity-product acceleration c
//   P2  lanes = (env, tendon):  crossing geometry, length, length rate, Hill force, wrench W per crossing
//   P4  lanes = (link, env, component):  pT = sum of the tendon wrenches on the link (owner gathers: no atomics)
//   P5  level by level from the leaves, octets:  own spatial inertia row and bias force, children,
//       U = I^A s, D, u, I^a, p^a
//   P6  level by level from the root, octets:  a, qdd
// Every accumulation is a gather by its owner lane in table order: results are bit-reproducible
// and independent of which wave / slot an env occupies.  The phases of a wave are ordered by
// wave_sync() alone (LDS instructions of one wave execute in issue order); a workgroup is a few
// waves sharing one LDS copy of the robot tables (staged behind the only barrier).
//
// The model is DESIGN.md §2; oracle/physics_np.py (Jacobian-sum M, RNE bias, dense solve) is what
// this is checked against; tests/proto/aba_world.py is the fp64 prototype of this formulation
// (agrees with the oracle to 3e-15).  Algorithmic HBM bytes per env step: 4 (4 n_q + n_t + 1).
// State layout in HBM for this kernel class: env-major rows q[n][n_q], qd[n][n_q], goal[n][n_q] (a wave's
// E envs are one contiguous run; the SoA planes of the env-per-lane kernels would cost a cache line per
// joint and wave: 6.4x the algorithmic bytes by PMC).
#pragma once
#include <hip/hip_runtime.h>

#include "env_common.hpp"
#include "philox.hpp"
#include "tree_build.hpp"

namespace rbt {

// ------------------------------------------------------------------ device
struct V3 { float x, y, z; };
__device__ __forceinline__ V3 operator+(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
__device__ __forceinline__ V3 operator-(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
__device__ __forceinline__ V3 operator*(V3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }
__device__ __forceinline__ float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ __forceinline__ V3 cross(V3 a, V3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
__device__ __forceinline__ V3 ld3(const float *p) { return {p[0], p[1], p[2]}; }
__device__ __forceinline__ void st3(float *p, V3 v) { p[0] = v.x; p[1] = v.y; p[2] = v.z; }
struct M3 { float m[9]; };   // row-major
__device__ __forceinline__ V3 mul(const M3 &a, V3 v) {
    return {a.m[0] * v.x + a.m[1] * v.y + a.m[2] * v.z, a.m[3] * v.x + a.m[4] * v.y + a.m[5] * v.z,
            a.m[6] * v.x + a.m[7] * v.y + a.m[8] * v.z};
}
__device__ __forceinline__ V3 mulT(const M3 &a, V3 v) {   // a^T v
    return {a.m[0] * v.x + a.m[3] * v.y + a.m[6] * v.z, a.m[1] * v.x + a.m[4] * v.y + a.m[7] * v.z,
            a.m[2] * v.x + a.m[5] * v.y + a.m[8] * v.z};
}
__device__ __forceinline__ M3 mul(const M3 &a, const M3 &b) {
    M3 o;
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c) o.m[3 * r + c] = a.m[3 * r] * b.m[c] + a.m[3 * r + 1] * b.m[3 + c] + a.m[3 * r + 2] * b.m[6 + c];
    return o;
}
// symmetric 3x3 as xx,yy,zz,xy,xz,yz
__device__ __forceinline__ V3 symmul(const float *s, V3 v) {
    return {s[0] * v.x + s[3] * v.y + s[4] * v.z, s[3] * v.x + s[1] * v.y + s[5] * v.z, s[4] * v.x + s[5] * v.y + s[2] * v.z};
}

// Phases of a wave exchange data through the wave's own LDS working set only.  LDS
// instructions of one wave execute in issue order, so a later ds_read sees an earlier
// ds_write of another lane of the same wave; all that is needed between phases is that
// the compiler keeps that order.
__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

template <int CTRL>
__device__ __forceinline__ float dpp(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, true));
}
// lanes 0,1,2 of a quad hold components 0,1,2 of a vector: component (r+1)%3 and (r+2)%3 of the own lane r.
// Lane 3 (idle, shadows component 2) reads what lane 2 reads, so that it stays an exact copy of lane 2 through
// every level: left to rotate onto itself its rows drift, grow from level to level of a long chain and reach
// inf after ~15 levels - and 0 * inf in an octet sum poisons the env (found by tests/test_random_robots_gpu.py).
__device__ __forceinline__ float rot1(float v) { return dpp<0x09>(v); }   // quad_perm [1,2,0,0]
__device__ __forceinline__ float rot2(float v) { return dpp<0x52>(v); }   // quad_perm [2,0,1,1]
// sum over the 8 lanes of an octet, result in every lane: quad_perm(1,0,3,2), quad_perm(2,3,0,1), row_half_mirror
__device__ __forceinline__ float sum8(float v) {
    v += dpp<0xB1>(v);
    v += dpp<0x4E>(v);
    v += dpp<0x141>(v);
    return v;
}

// item `it` of a phase with C items per env -> (env slot, item); E is small: a compare chain beats a division
template <int E>
__device__ __forceinline__ void split(int it, int C, int &e, int &x) {
    e = 0;
#pragma unroll
    for (int k = 1; k < E; ++k) e += (it >= k * C) ? 1 : 0;
    x = it - e * C;
}

// one wave's view of the staged tables and of its envs' working sets
struct Ctx {
    const TreeDev &t;
    const float *tab;     // LDS copy of the tables
    float *ws;            // LDS working sets of this wave: E blocks of t.ES floats
    int lane;
    __device__ __forceinline__ int ti(int off) const { return __float_as_int(tab[off]); }
    __device__ __forceinline__ float tf(int off) const { return tab[off]; }
    // 24-bit multiplies (one full-rate v_mad_u32_u24; a 32-bit v_mul_lo is a quarter-rate instruction)
    __device__ __forceinline__ float *env(int e) const { return ws + __mul24(e, t.ES); }
    __device__ __forceinline__ float *link(int e, int i) const { return ws + (__mul24(e, t.ES) + __mul24(i, LS)); }
};

// Copy the robot tables to LDS: 16-byte loads, issued in batches before their stores.
__device__ __forceinline__ void stage_tables(const TreeDev &g, float *lds_tab, int tid, int nthreads) {
    float4 *dst = reinterpret_cast<float4 *>(lds_tab);
    for (int base = 0; base < g.n_vec4; base += 4 * nthreads) {
        float4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int k = base + u * nthreads + tid;
            v[u] = k < g.n_vec4 ? g.g_words[k] : float4{0.0f, 0.0f, 0.0f, 0.0f};
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int k = base + u * nthreads + tid;
            if (k < g.n_vec4) dst[k] = v[u];
        }
    }
    __syncthreads();
}

// swap the halves of an octet: lane k gets the value of lane 4 + k and vice versa
// (row_half_mirror: i <-> 7 - i, then quad_perm [3,2,1,0])
__device__ __forceinline__ float swap_half(float v) { return dpp<0x1B>(dpp<0x141>(v)); }

// lane r of an octet
struct OctLane {
    int half, kk, n1, n2, row6;    // half: 0 angular / 1 linear; kk: own component (idle lanes shadow 2); n1, n2: the next two, cyclically
    float sgn;                     // -1 angular / +1 linear
    bool act;                      // owns a row
    __device__ __forceinline__ explicit OctLane(int r) {
        half = r >> 2;
        const int k = r & 3;
        act = k < 3;
        kk = act ? k : 2;
        n1 = kk == 2 ? 0 : kk + 1; n2 = kk == 0 ? 2 : kk - 1;
        row6 = 3 * half + kk;
        sgn = half ? 1.0f : -1.0f;
    }
};

// ---- level records ----
struct Rec1 { int i, par, flags; float ax[3], org[3]; };
__device__ __forceinline__ Rec1 load_rec1(const Ctx &c, int L, int x) {
    const float4 *p = reinterpret_cast<const float4 *>(c.tab + c.t.o_rec1 + __mul24((L << c.t.lw_shift) + x, REC1));
    const float4 a = p[0], b = p[1], d = p[2];
    Rec1 r;
    r.i = __float_as_int(a.x); r.par = __float_as_int(a.y); r.flags = __float_as_int(a.z);
    r.ax[0] = b.x; r.ax[1] = b.y; r.ax[2] = b.z; r.org[0] = b.w; r.org[1] = d.x; r.org[2] = d.y;
    return r;
}
// P5's record in two parts: the integers (prefetched one level ahead: they give the addresses) and the
// link's constants (loaded with the link's data at the top of the body)
struct Rec5 { int i, par, flags, n_ext, ext[4], es, xslot, q; };
__device__ __forceinline__ Rec5 load_rec5(const Ctx &c, int L, int x) {
    const int q = (L << c.t.lw_shift) + x;
    const float4 *p = reinterpret_cast<const float4 *>(c.tab + c.t.o_rec5 + __mul24(q, REC5));
    const float4 a = p[0], b = p[1], d = p[2];
    Rec5 r;
    r.i = __float_as_int(a.x); r.par = __float_as_int(a.y); r.flags = __float_as_int(a.z); r.n_ext = __float_as_int(a.w);
    r.ext[0] = __float_as_int(b.x); r.ext[1] = __float_as_int(b.y); r.ext[2] = __float_as_int(b.z); r.ext[3] = __float_as_int(b.w);
    r.es = __float_as_int(d.x); r.xslot = __float_as_int(d.y); r.q = q;
    return r;
}
struct Rec5c { float m, com[3], I6[6], arm, damp; };
__device__ __forceinline__ Rec5c load_rec5c(const Ctx &c, int q) {
    const float4 *p = reinterpret_cast<const float4 *>(c.tab + c.t.o_rec5 + __mul24(q, REC5) + 12);
    const float4 f = p[0], g = p[1], h = p[2];
    Rec5c r;
    r.m = f.x; r.com[0] = f.y; r.com[1] = f.z; r.com[2] = f.w;
    r.I6[0] = g.x; r.I6[1] = g.y; r.I6[2] = g.z; r.I6[3] = g.w; r.I6[4] = h.x; r.I6[5] = h.y; r.arm = h.z; r.damp = h.w;
    return r;
}

// ---- P1: frame, joint axis, spatial velocity, velocity-product acceleration of one link.
//      Every lane of the octet computes row kk of the frame / component kk of the vectors (both halves
//      alike, so the quad rotations work in each); the angular half stores R, p, z, c_ang, the linear
//      half w, vO, sl, c_lin.  `cy` carries (row of R, p, w, vO) down the chain. ----
struct Carry1 { V3 Rrow; float p, w, vo; };
// the joint's own rotation exp(q [axis]x) by columns: evaluated once per joint by the joint's lane (tree_accel), not
// by the eight lanes of its octet inside the level loop (a sincos and ~25 vector instructions per level there)
struct Rot { V3 c0, c1, c2; };
__device__ __forceinline__ Rot rodrigues(V3 ax, float q) {
    float sn, cs;
    __sincosf(q, &sn, &cs);
    const float oc = 1.0f - cs;        // I + sin K + (1 - cos) K^2
    Rot r;
    r.c0 = {1.0f - oc * (ax.y * ax.y + ax.z * ax.z), sn * ax.z + oc * ax.x * ax.y, -sn * ax.y + oc * ax.x * ax.z};
    r.c1 = {-sn * ax.z + oc * ax.x * ax.y, 1.0f - oc * (ax.x * ax.x + ax.z * ax.z), sn * ax.x + oc * ax.y * ax.z};
    r.c2 = {sn * ax.y + oc * ax.x * ax.z, -sn * ax.x + oc * ax.y * ax.z, 1.0f - oc * (ax.x * ax.x + ax.y * ax.y)};
    return r;
}
__device__ __forceinline__ Rot load_rot(const Ctx &c, int e, int i) {
    const float *p = c.env(e) + c.t.o_W + c.t.n_q + __mul24(i, ROT);
    return {ld3(p), ld3(p + 3), ld3(p + 6)};
}
__device__ __forceinline__ void p1_body(const Ctx &c, int e, const OctLane &o, const Rec1 &lr, const Rot &rot, float qdi, Carry1 &cy) {
    float *me = c.link(e, lr.i);
    V3 Rrow = cy.Rrow;
    float pp = cy.p, wp0 = cy.w, wp1 = rot1(cy.w), wp2 = rot2(cy.w), vop = cy.vo;
    if (!(lr.flags & F_INH)) {
        Rrow = {o.kk == 0 ? 1.0f : 0.0f, o.kk == 1 ? 1.0f : 0.0f, o.kk == 2 ? 1.0f : 0.0f};
        pp = 0.0f; wp0 = 0.0f; wp1 = 0.0f; wp2 = 0.0f; vop = 0.0f;
        if (lr.par >= 0) {
            const float *pa = c.link(e, lr.par);
            Rrow = ld3(pa + O_RP + 3 * o.kk);
            pp = pa[O_RP + 9 + o.kk]; wp0 = pa[O_V + o.kk]; wp1 = pa[O_V + o.n1]; wp2 = pa[O_V + o.n2]; vop = pa[O_V + 3 + o.kk];
        }
    }
    const V3 ax = {lr.ax[0], lr.ax[1], lr.ax[2]}, org = {lr.org[0], lr.org[1], lr.org[2]};
    const V3 c0 = rot.c0, c1 = rot.c1, c2 = rot.c2;
    const V3 Ri = {dot(Rrow, c0), dot(Rrow, c1), dot(Rrow, c2)};      // row kk of R_p rot
    const float p = pp + dot(Rrow, org);                              // component kk of p_i
    const float z = dot(Rrow, ax);                                    // ... of the joint axis
    const float p1 = rot1(p), p2 = rot2(p), z1 = rot1(z), z2 = rot2(z);
    const float sl = p1 * z2 - p2 * z1;                               // (p x z)[kk]
    const float w = wp0 + z * qdi, vo = vop + sl * qdi;
    const float ca = (wp1 * z2 - wp2 * z1) * qdi;                     // (w_p x z)[kk] qd   (w_i x z_i = w_p x z_i)
    const float w1 = rot1(w), w2 = rot2(w), sl1 = rot1(sl), sl2 = rot2(sl), vo1 = rot1(vo), vo2 = rot2(vo);
    const float cl = ((w1 * sl2 - w2 * sl1) + (vo1 * z2 - vo2 * z1)) * qdi;   // (w x sl + vO x z)[kk] qd
    if (o.act) {
        if (o.half == 0) {
            st3(me + O_RP + 3 * o.kk, Ri);
            me[O_RP + 9 + o.kk] = p; me[O_S + o.kk] = z; me[O_C + o.kk] = ca;
        } else {
            me[O_V + o.kk] = w; me[O_V + 3 + o.kk] = vo; me[O_S + 3 + o.kk] = sl; me[O_C + 3 + o.kk] = cl;
        }
    }
    cy.Rrow = Ri; cy.p = p; cy.w = w; cy.vo = vo;
}

// world position and velocity of a point fixed to `link` (local coordinates r); link < 0: the base
__device__ __forceinline__ void point_on_link(const Ctx &c, int e, int link, V3 r, V3 &x, V3 &xd) {
    if (link < 0) { x = r; xd = {0.0f, 0.0f, 0.0f}; return; }
    const float *lk = c.link(e, link);
    M3 R;
#pragma unroll
    for (int k = 0; k < 9; ++k) R.m[k] = lk[O_RP + k];
    x = ld3(lk + O_RP + 9) + mul(R, r);
    xd = ld3(lk + O_V + 3) + cross(ld3(lk + O_V), x);           // vO + w x x  (velocity about the world origin)
}

// ---- P2: one tendon: crossing geometry, Hill force, wrench of every crossing ----
__device__ __forceinline__ void p2_tendon(const Ctx &c, int e, int k) {
    const TreeDev &t = c.t;
    const int c0 = c.ti(t.o_t_cr_start + k), c1 = c.ti(t.o_t_cr_start + k + 1);
    float *W = c.env(e) + t.o_W;
    float len = 0.0f, ldot = 0.0f;
    // the last crossing's unit wrench stays in registers (a tendon with one crossing - every tendon of the upper
    // body - never reads its wrench back from LDS); earlier crossings are stored unscaled and rescaled below
    V3 wa = {0.0f, 0.0f, 0.0f}, wu = {0.0f, 0.0f, 0.0f};
    for (int cr = c0; cr < c1; ++cr) {
        const float *rec = c.tab + t.o_cross + cr * CROSS_REC;
        const int la = __float_as_int(rec[0]), lb = __float_as_int(rec[1]);
        V3 xa, va, xb, vb;
        point_on_link(c, e, la, ld3(rec + 2), xa, va);
        point_on_link(c, e, lb, ld3(rec + 5), xb, vb);
        const V3 d = xb - xa;
        const float d2 = dot(d, d), inv = __builtin_amdgcn_rsqf(d2);
        const V3 u = d * inv;
        len += d2 * inv;
        ldot += dot(u, vb - va);
        wa = cross(xa, u); wu = u;                                    // unit wrench; scaled by the tension below
        if (cr + 1 < c1) { st3(W + 6 * cr, wa); st3(W + 6 * cr + 3, wu); }
    }
    const float *tr = c.tab + t.o_tendon + k * TENDON_REC;
    const float es = len * tr[0] + tr[1];                             // scaled strain s (l/l0 - 1)
    const float act = __builtin_amdgcn_fmed3f(t.kps * es - (c.env(e) + t.o_SPU)[k], 0.0f, 1.0f);
    const float fl = __builtin_amdgcn_exp2f(-(es * es));
    const float v = ldot * tr[4];
    const float vp = fmaxf(v, 0.0f), p = __builtin_amdgcn_fmed3f(v + 1.0f, 0.0f, 1.0f);
    const float num = t.fv_c1l * vp + p, den = t.fv_c2l * vp + (t.fv_c2s * p + t.fv_k);
    const float fpe = fmaxf(__builtin_amdgcn_exp2f(t.pe_k2s * es) * t.inv_pe_den - t.inv_pe_den, 0.0f);
    const float F = tr[3] * ((act * fl) * num * __builtin_amdgcn_rcpf(den) + fpe);
    for (int cr = c0; cr + 1 < c1; ++cr) {
#pragma unroll
        for (int a = 0; a < 6; ++a) W[6 * cr + a] *= F;
    }
    if (c1 > c0) { st3(W + 6 * (c1 - 1), wa * F); st3(W + 6 * (c1 - 1) + 3, wu * F); }
}

// ---- P5: backward pass of one link.  Lane (half, kk) owns row row6 = 3 half + kk of the link's 6x6 and
//      component row6 of its 6-vectors; rows and vectors are held split in "own-half" and "other-half"
//      columns, each in rotated order (kk, kk+1, kk+2): rO[j] = I[row6][3 half + (kk+j)%3],
//      rX[j] = I[row6][3 (1 - half) + (kk+j)%3].  `cy` carries (rO, rX, p^a) up the chain. ----
struct Carry5 { float rO[3], rX[3], pa; };
__device__ __forceinline__ void p5_body(const Ctx &c, int e, const OctLane &o, const Rec5 &lr, Carry5 &cy) {
    const TreeDev &t = c.t;
    float *me = c.link(e, lr.i);
    const Rec5c lc = load_rec5c(c, lr.q);
    const int k = o.kk, n1 = o.n1, n2 = o.n2, hO = 3 * o.half, hX = 3 - hO;
    // own link's data in component order (k, n1, n2): the lane loads component k (three lane-dependent bases, the
    // fields at immediate offsets) and takes the other two from its quad by DPP rotations, the other half's
    // through a half swap - 11 LDS loads per link and lane instead of 33, each of which cost an address add
    const float *mO = me + hO + k, *mX = me + hX + k, *mk = me + k;
    const float sO0 = mO[O_S], sX0 = mX[O_S], cO0 = mO[O_C], cX0 = mX[O_C];
    const float sO[3] = {sO0, rot1(sO0), rot2(sO0)}, sX[3] = {sX0, rot1(sX0), rot2(sX0)};
    const float cO[3] = {cO0, rot1(cO0), rot2(cO0)}, cX[3] = {cX0, rot1(cX0), rot2(cX0)};
    const float pT = mO[O_PT];
    const float qdi = (c.env(e) + t.o_SQD)[lr.i];
    float rO[3] = {0.0f, 0.0f, 0.0f}, rX[3] = {0.0f, 0.0f, 0.0f};
    float pa = pT;                                         // the tendon wrenches on the link
    if (!(lr.flags & F_MASSLESS)) {                        // (whole levels of massless links take this uniformly)
        const V3 Rk = ld3(me + O_RP + 3 * k);
        const V3 R1 = {rot1(Rk.x), rot1(Rk.y), rot1(Rk.z)}, R2 = {rot2(Rk.x), rot2(Rk.y), rot2(Rk.z)};
        const float pk = mk[O_RP + 9];
        // spatial velocity in (own half, other half) order: for the angular lanes that is (w, vO), for the linear (vO, w)
        const float vO0 = mO[O_V], vX0 = mX[O_V];
        const float vO1 = rot1(vO0), vO2 = rot2(vO0), vX1 = rot1(vX0), vX2 = rot2(vX0);
        // spatial inertia of the link about the world origin, row row6
        const V3 com = {lc.com[0], lc.com[1], lc.com[2]};
        const float m = lc.m;
        const float ck = pk + dot(Rk, com), c1 = rot1(ck), c2 = rot2(ck);                        // world COM, rotated order
        const float hk = m * ck, h1 = m * c1, h2 = m * c2;
        const V3 tt = symmul(lc.I6, Rk);                   // I R_k^T  (I symmetric)
        const float Ikk = dot(tt, Rk) + m * (c1 * c1 + c2 * c2);
        const float Ik1 = dot(tt, R1) - hk * c1, Ik2 = dot(tt, R2) - hk * c2;
        // angular row k: [ Ibar row | [h]x row ] ;  linear row k: [ m e_k | -[h]x row ] ;  [h]x row k = (0, -h2, h1) in rotated order
        rO[0] = o.half ? m : Ikk; rO[1] = o.half ? 0.0f : Ik1; rO[2] = o.half ? 0.0f : Ik2;
        rX[1] = o.sgn * h2; rX[2] = -o.sgn * h1;
        // I v, row row6: the row times the velocity in the same (own, other) order - angular (Ibar w + h x vO)[k],
        // linear (m vO - h x w)[k] - without a branch on the half
        const float X = rO[0] * vO0 + rO[1] * vO1 + rO[2] * vO2 + rX[1] * vX1 + rX[2] * vX2;
        const float X1 = rot1(X), X2 = rot2(X);
        const float Y = swap_half(X), Y1 = rot1(Y), Y2 = rot2(Y);
        // bias force v x* (I v):  angular (w x Iv_a + vO x Iv_l)[k],  linear (w x Iv_l)[k]
        const float w1 = o.half ? vX1 : vO1, w2 = o.half ? vX2 : vO2;
        pa += (w1 * X2 - w2 * X1) + (o.half ? 0.0f : (vX1 * Y2 - vX2 * Y1));
    }
    // children: the one below in the same octet (registers), the others through their exchange slots
    if (lr.flags & F_REGCHILD) {
#pragma unroll
        for (int j = 0; j < 3; ++j) { rO[j] += cy.rO[j]; rX[j] += cy.rX[j]; }
        pa += cy.pa;
    }
    float *X0 = c.env(e) + t.o_W;
    auto add_ext = [&](int xs) {
        const float *row = X0 + __mul24(xs, XSLOT) + 6 * o.row6;
        rO[0] += row[hO + k]; rO[1] += row[hO + n1]; rO[2] += row[hO + n2];
        rX[0] += row[hX + k]; rX[1] += row[hX + n1]; rX[2] += row[hX + n2];
        pa += X0[__mul24(xs, XSLOT) + 36 + o.row6];
    };
    if (lr.n_ext > 0) {                                    // branch points only: one test on the chains
        add_ext(lr.ext[0]);
        if (lr.n_ext > 1) add_ext(lr.ext[1]);
        if (lr.n_ext > 2) add_ext(lr.ext[2]);
        if (lr.n_ext > 3) add_ext(lr.ext[3]);
        for (int q = 4; q < lr.n_ext; ++q) add_ext(c.ti(t.o_ext_list + lr.es + q));
    }
    const float U = rO[0] * sO[0] + rO[1] * sO[1] + rO[2] * sO[2] + rX[0] * sX[0] + rX[1] * sX[1] + rX[2] * sX[2];
    // s_row6 U_row6 and s_row6 p^a_row6; idle lanes add exact zeros to the sums (a select, not a product with 0)
    const float D = sum8(o.act ? sO[0] * U : 0.0f) + lc.arm;
    const float T = sum8(o.act ? sO[0] * pa : 0.0f);
    const float invD = __builtin_amdgcn_rcpf(D);
    const float u = -lc.damp * qdi - T;
    // U of the whole octet, in this lane's column order
    const float UO[3] = {U, rot1(U), rot2(U)};
    const float Us = swap_half(U);
    const float UX[3] = {Us, rot1(Us), rot2(Us)};
    const float K = U * invD;
    float acc = pa + U * (u * invD);
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        rO[j] -= K * UO[j]; rX[j] -= K * UX[j];           // I^a = I^A - U U^T / D
        acc += rO[j] * cO[j] + rX[j] * cX[j];              // p^a = p^A + I^a c + U u / D
    }
    // U, 1/D, u for the forward pass overwrite R (every lane has read what it needs from the block)
    if (o.act) me[O_U + o.row6] = U;
    if (o.act && o.row6 == 0) { me[O_D] = invD; me[O_D + 1] = u; }
    if ((lr.flags & F_XWRITE) && o.act) {
        float *row = X0 + __mul24(lr.xslot, XSLOT) + 6 * o.row6;
        row[hO + k] = rO[0]; row[hO + n1] = rO[1]; row[hO + n2] = rO[2];
        row[hX + k] = rX[0]; row[hX + n1] = rX[1]; row[hX + n2] = rX[2];
        X0[__mul24(lr.xslot, XSLOT) + 36 + o.row6] = acc;
    }
#pragma unroll
    for (int j = 0; j < 3; ++j) { cy.rO[j] = rO[j]; cy.rX[j] = rX[j]; }
    cy.pa = acc;
}

// ---- P6: forward pass of one link: component row6 of the spatial acceleration, qdd by an octet sum.
//      `a` carries the parent's acceleration down the chain. ----
struct Data6 { float cr, sr, U, invD, u; };
__device__ __forceinline__ Data6 load_data6(const Ctx &c, int e, const OctLane &o, int i) {
    const float *me = c.link(e, i);
    Data6 d;
    d.cr = me[O_C + o.row6]; d.sr = me[O_S + o.row6]; d.U = me[O_U + o.row6]; d.invD = me[O_D]; d.u = me[O_D + 1];
    return d;
}
__device__ __forceinline__ void p6_body(const Ctx &c, int e, const OctLane &o, const Rec1 &lr, const Data6 &d, float a0, float &a) {
    float *me = c.link(e, lr.i);
    float ap = a;
    if (!(lr.flags & F_INH)) ap = lr.par >= 0 ? c.link(e, lr.par)[O_A + o.row6] : a0;
    ap += d.cr;
    const float qdd = (d.u - sum8(o.act ? d.U * ap : 0.0f)) * d.invD;     // idle lanes add exact zeros
    a = ap + d.sr * qdd;
    if ((lr.flags & F_ASTORE) && o.act) me[O_A + o.row6] = a;
    if (o.act && o.row6 == 0) me[O_D + 2] = qdd;
}

// The three level sweeps.  Every link of a level gets an octet of lanes: slot = octet * 8 + r, octet ->
// (env, slot x of the level).  The first pass of a level (slot = lane) is the chained one: its record
// (and, where cheap, its data) for the NEXT level is requested before this level's work, and the carry
// registers hand data from level to level.  Robots so wide that a level needs several passes
// (E * lw * 8 > 64) have no inheriting links (tree_build), so their extra passes carry nothing; SP (single
// pass) instantiations leave those loops out (fewer instructions and registers: 35.7 -> 33.1 us on the upper body).
// The level loops are unrolled by two over a pair of record variables (A: even trips, B: odd trips), each fetched
// one level ahead of its use: written as "cur = nxt" at the end of a trip the hand-over cost ~22 register copies
// per level and sweep (a third of P1's vector instructions).
template <int E, bool SP>
__device__ __forceinline__ void sweep_p1(const Ctx &c) {
    const TreeDev &t = c.t;
    const int lw = 1 << t.lw_shift, n_slots = (E << t.lw_shift) * 8;
    const OctLane o(c.lane & 7);
    const int e0 = (c.lane >> 3) >> t.lw_shift, x0 = (c.lane >> 3) & (lw - 1);
    const bool first = c.lane < n_slots;
    const float *sqd = c.env(first ? e0 : 0) + t.o_SQD;
    Carry1 cy = {{0.0f, 0.0f, 0.0f}, 0.0f, 0.0f, 0.0f};
    struct Lv { Rec1 r; Rot rot; float qd; };
    auto fetch = [&](int L) {
        Lv v;
        v.r = load_rec1(c, L, x0);
        const int i = v.r.i >= 0 ? v.r.i : 0;
        v.rot = load_rot(c, first ? e0 : 0, i);
        v.qd = sqd[i];
        return v;
    };
    auto work = [&](int L, const Lv &v) {
        if (first && v.r.i >= 0) p1_body(c, e0, o, v.r, v.rot, v.qd, cy);
        for (int slot = c.lane + 64; !SP && slot < n_slots; slot += 64) {
            const int oc = slot >> 3, e = oc >> t.lw_shift;
            const Rec1 lr = load_rec1(c, L, oc & (lw - 1));
            Carry1 none = {{0.0f, 0.0f, 0.0f}, 0.0f, 0.0f, 0.0f};
            if (lr.i >= 0) p1_body(c, e, o, lr, load_rot(c, e, lr.i), (c.env(e) + t.o_SQD)[lr.i], none);
        }
        wave_sync();
    };
    const int n = t.n_levels;
    Lv A = fetch(0), B = A;
    for (int L = 0; L < n; L += 2) {
        if (L + 1 < n) B = fetch(L + 1);
        work(L, A);
        if (L + 1 >= n) break;
        if (L + 2 < n) A = fetch(L + 2);
        work(L + 1, B);
    }
}

template <int E, bool SP>
__device__ __forceinline__ void sweep_p5(const Ctx &c) {
    const TreeDev &t = c.t;
    const int lw = 1 << t.lw_shift, n_slots = (E << t.lw_shift) * 8;
    const OctLane o(c.lane & 7);
    const int e0 = (c.lane >> 3) >> t.lw_shift, x0 = (c.lane >> 3) & (lw - 1);
    const bool first = c.lane < n_slots;
    Carry5 cy = {{0.0f, 0.0f, 0.0f}, {0.0f, 0.0f, 0.0f}, 0.0f};
    auto work = [&](int L, const Rec5 &r) {
        if (first && r.i >= 0) p5_body(c, e0, o, r, cy);
        for (int slot = c.lane + 64; !SP && slot < n_slots; slot += 64) {
            const int oc = slot >> 3, e = oc >> t.lw_shift;
            const Rec5 lr = load_rec5(c, L, oc & (lw - 1));
            Carry5 none = {{0.0f, 0.0f, 0.0f}, {0.0f, 0.0f, 0.0f}, 0.0f};
            if (lr.i >= 0) p5_body(c, e, o, lr, none);
        }
        wave_sync();
    };
    Rec5 A = load_rec5(c, t.n_levels - 1, x0), B = A;
    for (int L = t.n_levels - 1; L >= 0; L -= 2) {
        if (L >= 1) B = load_rec5(c, L - 1, x0);
        work(L, A);
        if (L < 1) break;
        if (L >= 2) A = load_rec5(c, L - 2, x0);
        work(L - 1, B);
    }
}

template <int E, bool SP>
__device__ __forceinline__ void sweep_p6(const Ctx &c) {
    const TreeDev &t = c.t;
    const int lw = 1 << t.lw_shift, n_slots = (E << t.lw_shift) * 8;
    const OctLane o(c.lane & 7);
    const int e0 = (c.lane >> 3) >> t.lw_shift, x0 = (c.lane >> 3) & (lw - 1);
    const bool first = c.lane < n_slots;
    const float g0 = t.g[0], g1 = t.g[1], g2 = t.g[2];   // scalars first: a lane-indexed t.g[] becomes a global load from the kernarg
    const float a0 = o.half ? -(o.kk == 0 ? g0 : (o.kk == 1 ? g1 : g2)) : 0.0f;    // base: fictitious acceleration -g
    float a = 0.0f;
    struct Lv { Rec1 r; Data6 d; };
    auto fetch = [&](int L) {
        Lv v;
        v.r = load_rec1(c, L, x0);
        v.d = load_data6(c, first ? e0 : 0, o, v.r.i >= 0 ? v.r.i : 0);
        return v;
    };
    auto work = [&](int L, const Lv &v) {
        if (first && v.r.i >= 0) p6_body(c, e0, o, v.r, v.d, a0, a);
        for (int slot = c.lane + 64; !SP && slot < n_slots; slot += 64) {
            const int oc = slot >> 3, e = oc >> t.lw_shift;
            const Rec1 lr = load_rec1(c, L, oc & (lw - 1));
            float none = 0.0f;
            if (lr.i >= 0) p6_body(c, e, o, lr, load_data6(c, e, o, lr.i), a0, none);
        }
        wave_sync();
    };
    const int n = t.n_levels;
    Lv A = fetch(0), B = A;
    for (int L = 0; L < n; L += 2) {
        if (L + 1 < n) B = fetch(L + 1);
        work(L, A);
        if (L + 1 >= n) break;
        if (L + 2 < n) A = fetch(L + 2);
        work(L + 1, B);
    }
}

template <int E> struct Passes { static constexpr int N = (E * MAXQ + 63) / 64; };

// joint slot of pass p: (env slot, joint); false for padding slots
template <int E>
__device__ __forceinline__ bool joint_slot(const TreeDev &t, int lane, int p, int &e, int &j) {
    const int slot = lane + 64 * p;
    e = slot >> t.q_shift;
    j = slot & ((1 << t.q_shift) - 1);
    return e < E && j < t.n_q;
}

// qdd of the joints this lane owns (Passes<E>::N slots), from their q / qd
template <int E, bool SP>
__device__ __forceinline__ void tree_accel(const Ctx &c, const float *qj, const float *vj, float *qdd) {
    constexpr int NP = Passes<E>::N;
    const TreeDev &t = c.t;
    const int lane = c.lane;
#pragma unroll
    for (int p = 0; p < NP; ++p) {
        int e, j;
        if (joint_slot<E>(t, lane, p, e, j)) {
            // the joint's velocity and its rotation matrix (behind SQ at the start of W, dead before P2 writes W)
            const float *jr = c.tab + t.o_joint + __mul24(j, JOINT_REC);
            const Rot r = rodrigues({jr[4], jr[5], jr[6]}, qj[p]);
            float *dst = c.env(e) + t.o_W + t.n_q + __mul24(j, ROT);
            st3(dst, r.c0); st3(dst + 3, r.c1); st3(dst + 6, r.c2);
            (c.env(e) + t.o_SQD)[j] = vj[p];
        }
    }
    wave_sync();
    // ---- P1: forward kinematics, one tree level at a time ----
    if (!(RB_TREE_SKIP & 1)) sweep_p1<E, SP>(c);
    // the zero slot behind W that pads P4's lists (the rotation matrices or P5's exchange slots may have covered it;
    // P2 writes below it)
    if (lane < 6 * E) (c.env(lane / 6) + t.o_W + t.zoff)[lane % 6] = 0.0f;
    // ---- P2: tendons (SQ is dead: W overwrites it) ----
    for (int it = lane; it < ((RB_TREE_SKIP & 2) ? 0 : E * t.n_t); it += 64) {
        int e, k;
        split<E>(it, t.n_t, e, k);
        p2_tendon(c, e, k);
    }
    wave_sync();
    // ---- P4: pT = tendon wrenches on the link (enters the bias force); lanes = (link, env, component), link-major,
    //      lists sorted by falling length and padded to 4: every trip is 4 independent loads, added in list order ----
    for (int it = lane; it < ((RB_TREE_SKIP & 8) ? 0 : t.n_q * E * 6); it += 64) {
        const int t2 = it / 6, comp = it - 6 * t2;
        const int a = t2 / E, e = t2 - a * E;
        const int i = c.ti(t.o_lc_link + a);
        const float *W = c.env(e) + t.o_W + comp;
        float acc = 0.0f;
        const int s0 = c.ti(t.o_lc_start + a), s1 = c.ti(t.o_lc_start + a + 1);
        for (int idx = s0; idx < s1; idx += 4) {
            const int4 en = *reinterpret_cast<const int4 *>(c.tab + t.o_lc_list + idx);
            const float w0 = W[en.x >> 1], w1 = W[en.y >> 1], w2 = W[en.z >> 1], w3 = W[en.w >> 1];
            acc += (en.x & 1) ? w0 : -w0;
            acc += (en.y & 1) ? w1 : -w1;
            acc += (en.z & 1) ? w2 : -w2;
            acc += (en.w & 1) ? w3 : -w3;
        }
        c.link(e, i)[O_PT + comp] = acc;
    }
    wave_sync();
    // ---- P5: own inertia, children, articulated quantities, leaves to root ----
    if (!(RB_TREE_SKIP & 16)) sweep_p5<E, SP>(c);
    // ---- P6: accelerations, root to leaves ----
    if (!(RB_TREE_SKIP & 32)) sweep_p6<E, SP>(c);
#pragma unroll
    for (int p = 0; p < NP; ++p) {
        int e, j;
        qdd[p] = joint_slot<E>(t, lane, p, e, j) ? c.link(e, j)[O_D + 2] : 0.0f;
    }
    wave_sync();
}

// one env step of the joints this lane owns: n_substeps integrator substeps with the
// set-points held, velocity saturation and joint limits; ok[p] = false where a joint hit a limit
template <int INTEG, int E, bool SP>
__device__ __forceinline__ void tree_integrate(const Ctx &c, float *qj, float *vj, bool *ok) {
    constexpr int NP = Passes<E>::N;
    const TreeDev &t = c.t;
    float vmax[NP], lo[NP], hi[NP];
    bool joint[NP];
#pragma unroll
    for (int p = 0; p < NP; ++p) {
        int e, j;
        joint[p] = joint_slot<E>(t, c.lane, p, e, j);
        const float *rec = c.tab + t.o_joint + (joint[p] ? j : 0) * JOINT_REC;
        vmax[p] = joint[p] ? rec[2] : 0.0f; lo[p] = joint[p] ? rec[0] : 0.0f; hi[p] = joint[p] ? rec[1] : 0.0f;
        ok[p] = true;
    }
    const float h = t.h;
    auto sat = [&](float v, int p) { return __builtin_amdgcn_fmed3f(v, -vmax[p], vmax[p]); };
    for (int sub = 0; sub < t.nsub; ++sub) {
        if (INTEG == 0) {
            float a[NP];
            tree_accel<E, SP>(c, qj, vj, a);
#pragma unroll
            for (int p = 0; p < NP; ++p) { vj[p] = sat(vj[p] + h * a[p], p); qj[p] = qj[p] + h * vj[p]; }
        } else {
            // RK4 with every stage velocity saturated, as a loop over the four stages (one copy of the
            // acceleration code instead of four: a quarter of the instructions to fetch and far fewer
            // registers live across it); the weighted sums accumulate in the order k1 + 2 k2 + 2 k3 + k4
            const float h6 = h * (1.0f / 6.0f);
            float kq[NP], kv[NP], qs[NP], qa[NP], va[NP];
#pragma unroll
            for (int p = 0; p < NP; ++p) { kq[p] = sat(vj[p], p); qs[p] = qj[p]; qa[p] = 0.0f; va[p] = 0.0f; }
#pragma unroll 1
            for (int st = 0; st < 4; ++st) {
                tree_accel<E, SP>(c, qs, kq, kv);
                const float wgt = (st == 0 || st == 3) ? 1.0f : 2.0f, cst = st == 2 ? h : 0.5f * h;
#pragma unroll
                for (int p = 0; p < NP; ++p) {
                    qa[p] += wgt * kq[p]; va[p] += wgt * kv[p];
                    qs[p] = qj[p] + cst * kq[p];             // state of the next stage ...
                    kq[p] = sat(vj[p] + cst * kv[p], p);     // ... and its (saturated) velocity
                }
            }
#pragma unroll
            for (int p = 0; p < NP; ++p) { qj[p] = qj[p] + h6 * qa[p]; vj[p] = vj[p] + h6 * va[p]; }
        }
        // velocity saturation + joint limits
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            float v = sat(vj[p], p);
            const bool over = qj[p] > hi[p], under = qj[p] < lo[p];
            if (over) { qj[p] = hi[p]; v = fminf(v, 0.0f); }
            if (under) { qj[p] = lo[p]; v = fmaxf(v, 0.0f); }
            vj[p] = v;
            ok[p] = ok[p] && !(joint[p] && (over || under));
        }
    }
}

// per-env AND of the joints' flags: lanes publish into LDS, one lane per env combines.
// scratch: E * MAXQ floats at the start of env 0's W region (dead outside tree_accel) - any env block works
template <int E>
__device__ __forceinline__ bool env_all_ok(const Ctx &c, const bool *ok, int e_query) {
    constexpr int NP = Passes<E>::N;
    const TreeDev &t = c.t;
#pragma unroll
    for (int p = 0; p < NP; ++p) {
        int e, j;
        if (joint_slot<E>(t, c.lane, p, e, j)) (c.env(e) + t.o_W)[j] = ok[p] ? 1.0f : 0.0f;
    }
    wave_sync();
    bool all = true;
    if (e_query >= 0) {
        const float *f = c.env(e_query) + t.o_W;
        for (int j = 0; j < t.n_q; ++j) all = all && (f[j] != 0.0f);
    }
    wave_sync();
    return all;
}

#ifdef RB_TREE_DEBUG
// debug builds only (make ... FLAGS+=-DRB_TREE_DEBUG; rb_debug_tree_arm / rb_debug_tree_fetch in roboy_sim.hip):
// the wave that owns env rb_tree_dbg_env copies its LDS working set (E blocks of ES floats, as the last
// acceleration evaluation left them) to rb_tree_dbg_out
__device__ float *rb_tree_dbg_out = nullptr;
__device__ long rb_tree_dbg_env = -1;
#endif

template <int INTEG, int E, bool SP>
__global__ void __launch_bounds__(512, RB_TREE_MIN_WAVES)
tree_step_aba(const TreeDev tg, float *__restrict__ q, float *__restrict__ qd, uint32_t *__restrict__ feas,
              const float *__restrict__ act, float act_scale, long n) {
    constexpr int NP = Passes<E>::N;
    extern __shared__ float4 lds_raw4[];
    float *lds = reinterpret_cast<float *>(lds_raw4);
    stage_tables(tg, lds, threadIdx.x, blockDim.x);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const long env0 = (long(blockIdx.x) * nw + wave) * E;      // first env of this wave
    if (env0 >= n) return;                                      // whole wave idle (no barrier follows)
    const Ctx c{tg, lds, lds + 4 * tg.n_vec4 + wave * (E * tg.ES), lane};
    // a slot past the end of the batch shadows the last env and stores nothing
    float qj[NP], vj[NP];
    bool ok[NP];
#pragma unroll
    for (int p = 0; p < NP; ++p) {
        int e, j;
        const bool js = joint_slot<E>(tg, lane, p, e, j);
        const long env = env0 + e < n ? env0 + e : n - 1;
        qj[p] = js ? q[env * tg.n_q + j] : 0.0f;          // env-major rows [n][n_q]: a wave's E envs are contiguous
        vj[p] = js ? qd[env * tg.n_q + j] : 0.0f;
    }
    // activation offsets u = ksg * set-point (what the tendon phase consumes), once per env step
    for (int it = lane; it < E * tg.n_t; it += 64) {
        int e, k;
        split<E>(it, tg.n_t, e, k);
        const long env = env0 + e < n ? env0 + e : n - 1;
        (c.env(e) + tg.o_SPU)[k] = act[env * tg.n_t + k] * (act_scale * c.tf(tg.o_tendon + k * TENDON_REC + 2));
    }
    tree_integrate<INTEG, E, SP>(c, qj, vj, ok);
#ifdef RB_TREE_DEBUG
    if (rb_tree_dbg_out && rb_tree_dbg_env >= env0 && rb_tree_dbg_env < env0 + E)
        for (int k = lane; k < E * tg.ES; k += 64) rb_tree_dbg_out[k] = c.ws[k];
#endif
    const bool all_ok = env_all_ok<E>(c, ok, lane < E ? lane : -1);
#pragma unroll
    for (int p = 0; p < NP; ++p) {
        int e, j;
        if (joint_slot<E>(tg, lane, p, e, j) && env0 + e < n) { q[(env0 + e) * tg.n_q + j] = qj[p]; qd[(env0 + e) * tg.n_q + j] = vj[p]; }
    }
    if (lane < E && env0 + lane < n) feas[env0 + lane] = all_ok ? 1u : 0u;
}

// RoboyEnv.step fused around the tree step (the env layer of roboy_sim.hip's
// msj_env_step_kernel for joint-tree robots; same semantics): the lanes that
// own joint (e, j) rescale nothing themselves - lanes (e, k) rescale action k - hold their
// joint and its goal, publish (q - goal)^2 and qd^2 into LDS where lane e sums them in joint
// order; lane e evaluates reward / done, publishes the decision, and the joint lanes draw
// their own goal component on done.
template <int INTEG, int E, bool SP>
__global__ void __launch_bounds__(512, RB_TREE_MIN_WAVES)
tree_env_step_aba(const TreeDev tg, const rbe::EnvParams ep, const rbe::GoalBox box,
                  float *__restrict__ q, float *__restrict__ qd, uint32_t *__restrict__ feas,
                  float *__restrict__ goal, uint32_t *__restrict__ step_num, float *__restrict__ ep_ret,
                  uint32_t *__restrict__ goal_count, const float *__restrict__ act,
                  float *__restrict__ obs, float *__restrict__ reward, uint32_t *__restrict__ done,
                  double *__restrict__ ep_sum, uint32_t *__restrict__ ep_cnt, uint32_t *__restrict__ infeas_n,
                  long n, uint64_t seed, uint64_t env_id0) {
    constexpr int NP = Passes<E>::N;
    extern __shared__ float4 lds_raw4[];
    float *lds = reinterpret_cast<float *>(lds_raw4);
    stage_tables(tg, lds, threadIdx.x, blockDim.x);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const long env0 = (long(blockIdx.x) * nw + wave) * E;
    if (env0 >= n) return;
    const Ctx c{tg, lds, lds + 4 * tg.n_vec4 + wave * (E * tg.ES), lane};
    const int nq = tg.n_q;
    float qj[NP], vj[NP], gj[NP];
    bool ok[NP];
#pragma unroll
    for (int p = 0; p < NP; ++p) {
        int e, j;
        const bool js = joint_slot<E>(tg, lane, p, e, j);
        const long env = env0 + e < n ? env0 + e : n - 1;
        qj[p] = js ? q[env * tg.n_q + j] : 0.0f;          // env-major rows [n][n_q]: a wave's E envs are contiguous
        vj[p] = js ? qd[env * tg.n_q + j] : 0.0f;
        gj[p] = js ? goal[env * tg.n_q + j] : 0.0f;
    }
    for (int it = lane; it < E * tg.n_t; it += 64) {
        int e, k;
        split<E>(it, tg.n_t, e, k);
        const long env = env0 + e < n ? env0 + e : n - 1;
        // clamp to the action box, then slope * (x - in_high) + out_high with two roundings (roboy_env.py:157-158)
        const float x = fminf(fmaxf(act[env * tg.n_t + k], -1.0f), 1.0f);
        (c.env(e) + tg.o_SPU)[k] = rbe::mul_then_add(ep.slope, x - 1.0f, ep.act_hi) * c.tf(tg.o_tendon + k * TENDON_REC + 2);
    }
    tree_integrate<INTEG, E, SP>(c, qj, vj, ok);
    // publish the per-joint terms: W region, [0..nq) flags, [nq..2nq) dq^2, [2nq..3nq) qd^2  (6 n_cr >= ... not guaranteed:
    // use the link blocks instead - they are dead here: link j's slots 0, 1, 2)
#pragma unroll
    for (int p = 0; p < NP; ++p) {
        int e, j;
        if (joint_slot<E>(tg, lane, p, e, j)) {
            float *lk = c.link(e, j);
            const float dq = qj[p] - gj[p];
            lk[0] = ok[p] ? 1.0f : 0.0f; lk[1] = dq * dq; lk[2] = vj[p] * vj[p];
        }
    }
    wave_sync();
    // lane e (< E) is the env's accountant
    const bool acct = lane < E && env0 + lane < n;
    const long me = env0 + (lane < E ? lane : 0);
    bool all_ok = true, dn = false, reached = false;
    float r = 0.0f;
    uint32_t sn = 0u;
    if (acct) {
        float dq2 = 0.0f, dv2 = 0.0f;
        for (int j = 0; j < nq; ++j) {
            const float *lk = c.link(lane, j);
            all_ok = all_ok && (lk[0] != 0.0f); dq2 += lk[1]; dv2 += lk[2];
        }
        sn = step_num[me] + 1u;
        r = rbe::env_reward(ep, dq2, dv2, all_ok, reached);
        dn = reached || (sn > uint32_t(ep.max_len));
    }
    wave_sync();
    if (lane < E) c.link(lane, 0)[0] = (acct && dn) ? 1.0f : 0.0f;
    wave_sync();
    // joint lanes: observation, goal redraw on done, state write-back
#pragma unroll
    for (int p = 0; p < NP; ++p) {
        int e, j;
        if (!(joint_slot<E>(tg, lane, p, e, j) && env0 + e < n)) continue;
        const long env = env0 + e;
        const bool edn = c.link(e, 0)[0] != 0.0f;
        float oq = qj[p], ov = vj[p], og = gj[p];
        if (edn) {
            const uint64_t gid = env_id0 + uint64_t(env);
            const uint32_t draw = goal_count[env];         // read before the accountant lane bumps it (below, after a sync)
            auto draw_goal = [&](uint32_t d) {
                const rb::Philox4 rnd = rb::philox_draw(seed, gid, d, rb::STREAM_GOALS, uint32_t(j >> 2));
                return rbe::goal_value(box.lo[j], box.hi[j], rnd.v[j & 3]);
            };
            // RoboyEnv.step: _set_new_goal (:67-68); with the VecEnv worker's env.reset() (:82-87) a second draw replaces the first
            // unseen: only that one is evaluated (the counter still advances by two)
            gj[p] = draw_goal(draw + (ep.auto_reset ? 1u : 0u));
            if (ep.auto_reset) {
                qj[p] = 0.0f; vj[p] = 0.0f; oq = 0.0f; ov = 0.0f; og = gj[p];
            }
            goal[env * nq + j] = gj[p];
        }
        q[env * nq + j] = qj[p]; qd[env * nq + j] = vj[p];
        float *orow = obs + env * (3 * nq);
        orow[j] = oq; orow[nq + j] = ov; orow[2 * nq + j] = og;
    }
    wave_sync();
    if (acct) {
        float ret = ep_ret[me] + r;
        uint32_t fz = all_ok ? 1u : 0u;
        if (dn) {
            rbe::stat_add(&ep_sum[me], double(ret)); rbe::stat_add(&ep_sum[n + me], double(ret) * double(ret));
            rbe::stat_add(&ep_cnt[me], 1u); rbe::stat_add(&ep_cnt[n + me], sn - 1u); rbe::stat_add(&ep_cnt[2 * n + me], reached ? 1u : 0u);
            rbe::stat_add(&goal_count[me], ep.auto_reset ? 2u : 1u);
            if (ep.auto_reset) { sn = 1u; fz = 1u; }
            ret = 0.0f;
        }
        feas[me] = fz; step_num[me] = sn; ep_ret[me] = ret; reward[me] = r; done[me] = dn ? 1u : 0u;
        if (!all_ok) __hip_atomic_fetch_add(&infeas_n[me], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);     // (returns nothing: no load to wait for)
    }
}

// reset of the env layer for any n_q: zero pose, counter 1, a fresh goal, reset observation
__global__ void tree_env_reset_kernel(const rbe::GoalBox box, float *q, float *qd, uint32_t *feas, float *goal,
                                      uint32_t *step_num, float *ep_ret, uint32_t *goal_count, float *obs,
                                      int n_q, long n, uint64_t seed, uint64_t env0) {
    const long i = long(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t draw = goal_count[i];
    goal_count[i] = draw + 1u;
    for (int b = 0; 4 * b < n_q; ++b) {
        const rb::Philox4 r = rb::philox_draw(seed, env0 + uint64_t(i), draw, rb::STREAM_GOALS, uint32_t(b));
        for (int k = 0; k < 4 && 4 * b + k < n_q; ++k) {
            const int j = 4 * b + k;
            const float g = rbe::goal_value(box.lo[j], box.hi[j], r.v[k]);
            q[i * n_q + j] = 0.0f; qd[i * n_q + j] = 0.0f; goal[i * n_q + j] = g;
            if (obs) { obs[i * 3 * n_q + j] = 0.0f; obs[i * 3 * n_q + n_q + j] = 0.0f; obs[i * 3 * n_q + 2 * n_q + j] = g; }
        }
    }
    feas[i] = 1u; step_num[i] = 1u; ep_ret[i] = 0.0f;
}

}  // namespace rbt
