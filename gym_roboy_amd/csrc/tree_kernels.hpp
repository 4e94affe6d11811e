// tree_kernels.hpp - generic joint-tree robots (roboy-tendon-robot/1), one env per WAVE.
//
// For robots outside the ball-joint class (msj_math.hpp) - e.g. the 20-DOF /
// 38-tendon upper body of BASELINE.json configs[3] - a per-lane formulation
// does not fit: the 20x20 mass matrix alone is 400 floats per env.  Here the 64
// lanes of one wavefront cooperate on ONE env, with the env's working set
// (link frames, velocities, wrenches, composite inertias, M) in LDS:
//
//   lanes = links     forward kinematics + velocities + Newton-Euler forward
//                     pass, level by level down the tree
//   lanes = tendons   via-point routing, length, length rate, Hill force,
//                     force on every via-point
//   lanes = links     gather the via-point forces of the own link, then the
//                     backward pass (wrenches and composite inertias) level by
//                     level up the tree; generalized bias+tendon force
//   lanes = M entries composite-rigid-body mass matrix
//   wave              dense Cholesky in LDS, two triangular solves
//   lanes = joints    integrator, velocity/joint limits
//
// No atomics anywhere: every accumulation is a gather by its owner lane, so
// results are bit-reproducible.  A workgroup is TREE_WAVES waves = TREE_WAVES envs
// that share one LDS copy of the robot tables (staged once, behind the only
// workgroup barrier); each wave then works in its own LDS working set, and the
// phases of a wave are ordered by wave_sync() alone (LDS instructions of one
// wave execute in issue order).  Algorithmic HBM bytes per env step:
// 4*(4 n_q + n_t + 1).
//
// The model is the one of DESIGN.md §2; the oracle evaluates it with the
// textbook Jacobian-sum M and is what this file is checked against.
#pragma once
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <string>
#include <vector>

#include "../../include/roboy_sim.h"
#include "env_common.hpp"
#include "philox.hpp"

namespace rbt {

constexpr int MAXQ = 32;    // joints per robot (one lane each; anc_mask is 32 bits)
constexpr int MAXT = 64;    // tendons per robot (one lane each)
constexpr int MAXVP = 1024; // via-points

// ------------------------------------------------------------------ tables
struct TreeDev {
    int n_q, n_t, n_vp, n_levels, nsub;
    float h;
    float g[3];
    float kp, fl_k2, pe_k2, inv_pe_den, fv_c1l, fv_c2l, fv_c2s;
    // int tables
    const int *parent, *anc_mask, *order, *level_start, *child_start, *child_list, *lvp_start, *lvp_list,
        *vp_link, *t_first, *t_count, *pair_ij, *pair_start;
    // float tables
    const float *axis, *origin, *mass, *com, *inertia, *armature, *damping, *qlo, *qhi, *qdmax, *vp_pos,
        *t_inv_l0, *t_sg_l0, *t_fmax, *t_inv_vl0;
    // all tables as one device buffer of 32-bit words ([ints | floats], padded to 16 bytes):
    // a workgroup copies it to LDS once, shared by its TREE_WAVES envs
    const float4 *g_words;
    int n_vec4;               // 16-byte chunks in g_words
    int io[13], fo[15];       // word offset of each table in the buffer
};

constexpr int TREE_WAVES = 4;   // envs (= waves) per workgroup, sharing one LDS copy of the tables

struct TreeHost {
    std::vector<int> ints;
    std::vector<float> floats;
    // offsets into the two buffers, same order as the TreeDev pointers
    size_t io[13], fo[15];
    TreeDev dev;   // scalars filled; pointers patched after upload
    size_t lds_floats = 0;
};

// words of the combined [ints | floats] table buffer, padded to a 16-byte multiple
inline size_t tree_table_words(const TreeHost &h) { return (h.ints.size() + h.floats.size() + 3) / 4 * 4; }

inline size_t tree_lds_floats(int nq, int nvp) {
    // R 9, P Z W VP AL AP FF NN CH FK FL VO 3 each (12*3), CM 1, CI 6, SQ SQD RHS 3
    return size_t(nq) * (9 + 36 + 1 + 6 + 3) + size_t(nvp) * 3 + size_t(nq) * nq;
}

// Flatten the description; rest lengths and every derived constant in fp64.
inline int tree_build(const rb_robot_desc *d, double step_size, int nsub, TreeHost &out, std::string &err) {
    const int nq = d->n_q, nt = d->n_t, nvp = d->n_vp;
    if (nq < 1 || nq > MAXQ) { err = "generic-tree kernel supports 1..32 joints"; return RB_EUNSUPPORTED; }
    if (nt < 1 || nt > MAXT) { err = "generic-tree kernel supports 1..64 tendons"; return RB_EUNSUPPORTED; }
    if (nvp > MAXVP) { err = "too many via-points"; return RB_EUNSUPPORTED; }
    std::vector<int> level(nq), parent(d->parent, d->parent + nq), anc(nq, 0);
    int nlev = 0;
    for (int i = 0; i < nq; ++i) {
        if (parent[i] < -1 || parent[i] >= i) { err = "parent must be -1 or an earlier joint"; return RB_EINVAL; }
        level[i] = parent[i] < 0 ? 0 : level[parent[i]] + 1;
        anc[i] = (parent[i] < 0 ? 0 : anc[parent[i]]) | (1 << i);
        nlev = level[i] + 1 > nlev ? level[i] + 1 : nlev;
    }
    std::vector<int> order, level_start(nlev + 1, 0), child_start(nq + 1, 0), child_list, lvp_start(nq + 1, 0), lvp_list;
    for (int L = 0; L < nlev; ++L) {
        level_start[L] = int(order.size());
        for (int i = 0; i < nq; ++i) if (level[i] == L) order.push_back(i);
    }
    level_start[nlev] = nq;
    for (int i = 0; i < nq; ++i) {
        child_start[i] = int(child_list.size());
        for (int c = 0; c < nq; ++c) if (parent[c] == i) child_list.push_back(c);
        lvp_start[i] = int(lvp_list.size());
        for (int v = 0; v < nvp; ++v) if (d->vp_link[v] == i) lvp_list.push_back(v);
    }
    child_start[nq] = int(child_list.size());
    lvp_start[nq] = int(lvp_list.size());
    if (child_list.empty()) child_list.push_back(0);
    if (lvp_list.empty()) lvp_list.push_back(0);
    // rest lengths: zero pose => every link frame is a pure translation
    std::vector<double> org(3 * nq);
    for (int i = 0; i < nq; ++i)
        for (int a = 0; a < 3; ++a) org[3 * i + a] = (parent[i] < 0 ? 0.0 : org[3 * parent[i] + a]) + d->origin[3 * i + a];
    std::vector<int> t_first(nt), t_count(nt);
    std::vector<float> inv_l0(nt), sg_l0(nt), fmax(nt), inv_vl0(nt);
    for (int k = 0; k < nt; ++k) {
        const int v0 = d->vp_offset[k], v1 = d->vp_offset[k + 1];
        if (v1 - v0 < 2) { err = "tendon with fewer than two via-points"; return RB_EINVAL; }
        double l0 = 0.0;
        for (int v = v0; v + 1 < v1; ++v) {
            double s = 0.0;
            for (int a = 0; a < 3; ++a) {
                const int la = d->vp_link[v], lb = d->vp_link[v + 1];
                const double xa = (la < 0 ? 0.0 : org[3 * la + a]) + d->vp_pos[3 * v + a];
                const double xb = (lb < 0 ? 0.0 : org[3 * lb + a]) + d->vp_pos[3 * (v + 1) + a];
                s += (xb - xa) * (xb - xa);
            }
            if (s < 1e-12) { err = "degenerate tendon segment"; return RB_EINVAL; }
            l0 += std::sqrt(s);
        }
        t_first[k] = v0; t_count[k] = v1 - v0;
        inv_l0[k] = float(1.0 / l0); sg_l0[k] = float(d->setpoint_scale / l0);
        fmax[k] = float(d->f_max[k]); inv_vl0[k] = float(1.0 / (d->v_max * l0));
    }
    auto push_i = [&](int slot, const std::vector<int> &v) { out.io[slot] = out.ints.size(); out.ints.insert(out.ints.end(), v.begin(), v.end()); };
    auto push_d = [&](int slot, const double *p, size_t n) { out.fo[slot] = out.floats.size(); for (size_t i = 0; i < n; ++i) out.floats.push_back(float(p[i])); };
    auto push_f = [&](int slot, const std::vector<float> &v) { out.fo[slot] = out.floats.size(); out.floats.insert(out.floats.end(), v.begin(), v.end()); };
    out.ints.clear(); out.floats.clear();
    push_i(0, parent); push_i(1, anc); push_i(2, order); push_i(3, level_start); push_i(4, child_start);
    push_i(5, child_list); push_i(6, lvp_start); push_i(7, lvp_list);
    push_i(8, std::vector<int>(d->vp_link, d->vp_link + nvp)); push_i(9, t_first); push_i(10, t_count);
    // strictly-lower-triangle pairs (i > j) ordered by column j, then row i: the
    // trailing block of Cholesky column c is the contiguous range from pair_start[c + 1]
    std::vector<int> pair_ij, pair_start(nq + 1, 0);
    for (int j = 0; j < nq; ++j) {
        pair_start[j] = int(pair_ij.size());
        for (int i = j; i < nq; ++i) pair_ij.push_back((i << 8) | j);   // diagonal included (i == j)
    }
    pair_start[nq] = int(pair_ij.size());
    push_i(11, pair_ij); push_i(12, pair_start);
    push_d(0, d->axis, 3 * nq); push_d(1, d->origin, 3 * nq); push_d(2, d->mass, nq); push_d(3, d->com, 3 * nq);
    push_d(4, d->inertia, 6 * nq); push_d(5, d->armature, nq); push_d(6, d->damping, nq); push_d(7, d->q_lo, nq);
    push_d(8, d->q_hi, nq); push_d(9, d->qd_max, nq); push_d(10, d->vp_pos, 3 * nvp);
    push_f(11, inv_l0); push_f(12, sg_l0); push_f(13, fmax); push_f(14, inv_vl0);
    TreeDev &t = out.dev;
    t.n_q = nq; t.n_t = nt; t.n_vp = nvp; t.n_levels = nlev; t.nsub = nsub; t.h = float(step_size / nsub);
    for (int a = 0; a < 3; ++a) t.g[a] = float(d->gravity[a]);
    const double log2e = 1.4426950408889634;
    t.kp = float(d->kp);
    t.fl_k2 = float(-log2e / (d->fl_width * d->fl_width));
    t.pe_k2 = float(log2e * d->kpe / d->e0);
    t.inv_pe_den = float(1.0 / (std::exp(d->kpe) - 1.0));
    const double c2l = (1.0 + 1.0 / d->fv_a) / (d->fv_n - 1.0);
    t.fv_c2s = float(-1.0 / d->fv_a); t.fv_c1l = float(d->fv_n * c2l); t.fv_c2l = float(c2l);
    // per workgroup: one table copy + TREE_WAVES working sets
    out.lds_floats = tree_table_words(out) + size_t(TREE_WAVES) * tree_lds_floats(nq, nvp);
    return RB_OK;
}

__host__ __device__ inline int tree_lds_floats_dev(int nq, int nvp) { return nq * (9 + 36 + 1 + 6 + 3) + nvp * 3 + nq * nq; }

inline void tree_patch_pointers(TreeHost &h, const int *d_ints, const float *d_floats) {
    TreeDev &t = h.dev;
    const int **ip[13] = {&t.parent, &t.anc_mask, &t.order, &t.level_start, &t.child_start, &t.child_list,
                          &t.lvp_start, &t.lvp_list, &t.vp_link, &t.t_first, &t.t_count, &t.pair_ij, &t.pair_start};
    const float **fp[15] = {&t.axis, &t.origin, &t.mass, &t.com, &t.inertia, &t.armature, &t.damping, &t.qlo,
                            &t.qhi, &t.qdmax, &t.vp_pos, &t.t_inv_l0, &t.t_sg_l0, &t.t_fmax, &t.t_inv_vl0};
    for (int i = 0; i < 13; ++i) { *ip[i] = d_ints + h.io[i]; t.io[i] = int(h.io[i]); }
    for (int i = 0; i < 15; ++i) { *fp[i] = d_floats + h.fo[i]; t.fo[i] = int(h.ints.size() + h.fo[i]); }
    t.g_words = reinterpret_cast<const float4 *>(d_ints);   // the floats follow the ints in the same allocation
    t.n_vec4 = int(tree_table_words(h) / 4);
}

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

struct Lds {
    float *R, *P, *Z, *W, *VP, *AL, *AP, *FF, *NN, *CH, *FK, *FL, *VO, *CM, *CI, *SQ, *SQD, *RHS, *FV, *M;
    __device__ __forceinline__ Lds(float *b, int nq, int nvp) {
        R = b; b += 9 * nq;
        P = b; b += 3 * nq; Z = b; b += 3 * nq; W = b; b += 3 * nq; VP = b; b += 3 * nq; AL = b; b += 3 * nq;
        AP = b; b += 3 * nq; FF = b; b += 3 * nq; NN = b; b += 3 * nq; CH = b; b += 3 * nq; FK = b; b += 3 * nq;
        FL = b; b += 3 * nq; VO = b; b += 3 * nq;
        CM = b; b += nq; CI = b; b += 6 * nq; SQ = b; b += nq; SQD = b; b += nq; RHS = b; b += nq;
        FV = b; b += 3 * nvp; M = b;
    }
};

// world position (and velocity) of via-point v
__device__ __forceinline__ void via_point(const TreeDev &t, const Lds &s, int v, V3 &x, V3 &xd) {
    const int link = t.vp_link[v];
    const V3 pos = ld3(t.vp_pos + 3 * v);
    if (link < 0) { x = pos; xd = {0.0f, 0.0f, 0.0f}; return; }
    M3 R;
#pragma unroll
    for (int a = 0; a < 9; ++a) R.m[a] = s.R[9 * link + a];
    const V3 r = mul(R, pos);
    x = ld3(s.P + 3 * link) + r;
    xd = ld3(s.VP + 3 * link) + cross(ld3(s.W + 3 * link), r);
}

// Phases of one env exchange data through the wave's own LDS working set only.
// LDS instructions of one wave execute in issue order, so a later ds_read sees an
// earlier ds_write of another lane of the same wave; all that is needed between
// phases is that the compiler keeps that order.
__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Copy the robot tables to LDS (16-byte loads, issued in batches before their
// stores) and return a TreeDev whose tables point at the copy: later table reads
// are ds_reads instead of dependent global loads (257 per wave and evaluation
// before; they were the largest share of the wave's s_waitcnt time).
__device__ __forceinline__ TreeDev stage_tables(const TreeDev &g, float *lds_tab, int tid, int nthreads) {
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
    TreeDev t = g;
    const int *li = reinterpret_cast<const int *>(lds_tab);
    const float *lf = lds_tab;
    t.parent = li + g.io[0]; t.anc_mask = li + g.io[1]; t.order = li + g.io[2]; t.level_start = li + g.io[3];
    t.child_start = li + g.io[4]; t.child_list = li + g.io[5]; t.lvp_start = li + g.io[6]; t.lvp_list = li + g.io[7];
    t.vp_link = li + g.io[8]; t.t_first = li + g.io[9]; t.t_count = li + g.io[10]; t.pair_ij = li + g.io[11];
    t.pair_start = li + g.io[12];
    t.axis = lf + g.fo[0]; t.origin = lf + g.fo[1]; t.mass = lf + g.fo[2]; t.com = lf + g.fo[3]; t.inertia = lf + g.fo[4];
    t.armature = lf + g.fo[5]; t.damping = lf + g.fo[6]; t.qlo = lf + g.fo[7]; t.qhi = lf + g.fo[8]; t.qdmax = lf + g.fo[9];
    t.vp_pos = lf + g.fo[10]; t.t_inv_l0 = lf + g.fo[11]; t.t_sg_l0 = lf + g.fo[12]; t.t_fmax = lf + g.fo[13];
    t.t_inv_vl0 = lf + g.fo[14];
    return t;
}

// qdd of the own joint (lane < n_q); spk = set-point of the own tendon (lane < n_t)
__device__ __forceinline__ float tree_accel(const TreeDev &t, const Lds &s, int lane, float qj, float vj, float spk) {
    const int nq = t.n_q, nt = t.n_t;
    if (lane < nq) { s.SQ[lane] = qj; s.SQD[lane] = vj; }
    wave_sync();
    // ---- forward pass, one tree level at a time ----
    for (int L = 0; L < t.n_levels; ++L) {
        const int a = t.level_start[L], b = t.level_start[L + 1];
        if (lane >= a && lane < b) {
            const int i = t.order[lane];
            const int par = t.parent[i];
            M3 Rp = {{1, 0, 0, 0, 1, 0, 0, 0, 1}};
            V3 pp = {0, 0, 0}, wp = {0, 0, 0}, vpp = {0, 0, 0}, alp = {0, 0, 0};
            V3 app = {-t.g[0], -t.g[1], -t.g[2]};       // base acceleration -g
            if (par >= 0) {
#pragma unroll
                for (int e = 0; e < 9; ++e) Rp.m[e] = s.R[9 * par + e];
                pp = ld3(s.P + 3 * par); wp = ld3(s.W + 3 * par); vpp = ld3(s.VP + 3 * par);
                alp = ld3(s.AL + 3 * par); app = ld3(s.AP + 3 * par);
            }
            const V3 ax = ld3(t.axis + 3 * i);
            const V3 r = mul(Rp, ld3(t.origin + 3 * i));
            const V3 pi = pp + r;
            const V3 zi = mul(Rp, ax);
            float sn, cs;
            __sincosf(s.SQ[i], &sn, &cs);
            // Rodrigues: I + sin K + (1 - cos) K^2
            const float oc = 1.0f - cs;
            const M3 rot = {{1.0f - oc * (ax.y * ax.y + ax.z * ax.z), -sn * ax.z + oc * ax.x * ax.y, sn * ax.y + oc * ax.x * ax.z,
                             sn * ax.z + oc * ax.x * ax.y, 1.0f - oc * (ax.x * ax.x + ax.z * ax.z), -sn * ax.x + oc * ax.y * ax.z,
                             -sn * ax.y + oc * ax.x * ax.z, sn * ax.x + oc * ax.y * ax.z, 1.0f - oc * (ax.x * ax.x + ax.y * ax.y)}};
            const M3 Ri = mul(Rp, rot);
            const float qd = s.SQD[i];
            const V3 vpi = vpp + cross(wp, r);
            const V3 api = app + cross(alp, r) + cross(wp, cross(wp, r));
            const V3 wi = wp + zi * qd;
            const V3 ali = alp + cross(wp, zi) * qd;
            const float m = t.mass[i];
            const float *I6 = t.inertia + 6 * i;
#pragma unroll
            for (int e = 0; e < 9; ++e) s.R[9 * i + e] = Ri.m[e];
            st3(s.P + 3 * i, pi); st3(s.Z + 3 * i, zi); st3(s.W + 3 * i, wi); st3(s.VP + 3 * i, vpi);
            st3(s.AL + 3 * i, ali); st3(s.AP + 3 * i, api);
            float *ci = s.CI + 6 * i;
            // massless virtual links (the x/y joints of a ball joint) skip the
            // inertia arithmetic; whole levels of them take the branch uniformly
            if (m != 0.0f || I6[0] != 0.0f || I6[1] != 0.0f || I6[2] != 0.0f) {
                const V3 rc = mul(Ri, ld3(t.com + 3 * i));
                const V3 ac = api + cross(ali, rc) + cross(wi, cross(wi, rc));
                const V3 F = ac * m;
                // Iw = R I R^T (symmetric)
                M3 RI;
#pragma unroll
                for (int rr = 0; rr < 3; ++rr) {
                    const V3 row = {Ri.m[3 * rr], Ri.m[3 * rr + 1], Ri.m[3 * rr + 2]};
                    const V3 ri = symmul(I6, row);     // (R I)_row = I row (I symmetric)
                    RI.m[3 * rr] = ri.x; RI.m[3 * rr + 1] = ri.y; RI.m[3 * rr + 2] = ri.z;
                }
                float Iw[6];
                {
                    auto rowdot = [&](int r1, int r2) {
                        return RI.m[3 * r1] * Ri.m[3 * r2] + RI.m[3 * r1 + 1] * Ri.m[3 * r2 + 1] + RI.m[3 * r1 + 2] * Ri.m[3 * r2 + 2];
                    };
                    Iw[0] = rowdot(0, 0); Iw[1] = rowdot(1, 1); Iw[2] = rowdot(2, 2);
                    Iw[3] = rowdot(0, 1); Iw[4] = rowdot(0, 2); Iw[5] = rowdot(1, 2);
                }
                const V3 N = symmul(Iw, ali) + cross(wi, symmul(Iw, wi));
                st3(s.FF + 3 * i, F); st3(s.NN + 3 * i, N + cross(rc, F));
                // composite-inertia seed, about the world origin
                const V3 cw = pi + rc;
                const float c2 = dot(cw, cw);
                s.CM[i] = m; st3(s.CH + 3 * i, cw * m);
                ci[0] = Iw[0] + m * (c2 - cw.x * cw.x); ci[1] = Iw[1] + m * (c2 - cw.y * cw.y); ci[2] = Iw[2] + m * (c2 - cw.z * cw.z);
                ci[3] = Iw[3] - m * cw.x * cw.y; ci[4] = Iw[4] - m * cw.x * cw.z; ci[5] = Iw[5] - m * cw.y * cw.z;
            } else {
                const V3 zero = {0.0f, 0.0f, 0.0f};
                st3(s.FF + 3 * i, zero); st3(s.NN + 3 * i, zero); st3(s.CH + 3 * i, zero);
                s.CM[i] = 0.0f;
#pragma unroll
                for (int e = 0; e < 6; ++e) ci[e] = 0.0f;
            }
        }
        wave_sync();
    }
    // ---- tendons: one lane each ----
    if (lane < nt) {
        const int first = t.t_first[lane], cnt = t.t_count[lane];
        float len = 0.0f, ldot = 0.0f;
        V3 xa, va, xb, vb;
        via_point(t, s, first, xa, va);
        for (int k = 1; k < cnt; ++k) {
            via_point(t, s, first + k, xb, vb);
            const V3 d = xb - xa;
            const float d2 = dot(d, d), inv = __builtin_amdgcn_rsqf(d2);
            len += d2 * inv;
            ldot += dot(d, vb - va) * inv;
            xa = xb; va = vb;
        }
        const float e = len * t.t_inv_l0[lane] - 1.0f;
        const float act = fminf(fmaxf(t.kp * (e - t.t_sg_l0[lane] * spk), 0.0f), 1.0f);
        const float fl = __builtin_amdgcn_exp2f(t.fl_k2 * (e * e));
        const float v = ldot * t.t_inv_vl0[lane];
        const float vp = fmaxf(v, 0.0f), vm = fminf(fmaxf(v, -1.0f), 0.0f);
        const float num = t.fv_c1l * vp + (1.0f + vm), den = t.fv_c2l * vp + (t.fv_c2s * vm + 1.0f);
        const float fv = num * __builtin_amdgcn_rcpf(den);
        const float fpe = fmaxf((__builtin_amdgcn_exp2f(t.pe_k2 * e) - 1.0f) * t.inv_pe_den, 0.0f);
        const float F = t.t_fmax[lane] * (act * fl * fv + fpe);
        // force on every via-point: F (u_next - u_prev)
        V3 uprev = {0, 0, 0};
        via_point(t, s, first, xa, va);
        for (int k = 1; k < cnt; ++k) {
            via_point(t, s, first + k, xb, vb);
            const V3 d = xb - xa;
            const V3 u = d * __builtin_amdgcn_rsqf(dot(d, d));
            st3(s.FV + 3 * (first + k - 1), (u - uprev) * F);
            uprev = u; xa = xb;
        }
        st3(s.FV + 3 * (first + cnt - 1), uprev * (-F));
    }
    wave_sync();
    // ---- links gather the tendon forces applied to them (subtracted: the
    //      backward pass then yields bias - tendon generalized force) ----
    if (lane < nq) {
        const int i = lane;
        M3 R;
#pragma unroll
        for (int e = 0; e < 9; ++e) R.m[e] = s.R[9 * i + e];
        V3 f = ld3(s.FF + 3 * i), n = ld3(s.NN + 3 * i);
        for (int idx = t.lvp_start[i]; idx < t.lvp_start[i + 1]; ++idx) {
            const int v = t.lvp_list[idx];
            const V3 fv = ld3(s.FV + 3 * v);
            f = f - fv;
            n = n - cross(mul(R, ld3(t.vp_pos + 3 * v)), fv);
        }
        st3(s.FF + 3 * i, f); st3(s.NN + 3 * i, n);
    }
    wave_sync();
    // ---- backward pass: parents gather wrenches and composite inertias ----
    for (int L = t.n_levels - 2; L >= 0; --L) {
        const int a = t.level_start[L], b = t.level_start[L + 1];
        if (lane >= a && lane < b) {
            const int i = t.order[lane];
            V3 f = ld3(s.FF + 3 * i), n = ld3(s.NN + 3 * i), ch = ld3(s.CH + 3 * i);
            const V3 pi = ld3(s.P + 3 * i);
            float cm = s.CM[i], ci[6];
#pragma unroll
            for (int e = 0; e < 6; ++e) ci[e] = s.CI[6 * i + e];
            for (int idx = t.child_start[i]; idx < t.child_start[i + 1]; ++idx) {
                const int c = t.child_list[idx];
                const V3 fc = ld3(s.FF + 3 * c);
                f = f + fc;
                n = n + ld3(s.NN + 3 * c) + cross(ld3(s.P + 3 * c) - pi, fc);
                cm += s.CM[c]; ch = ch + ld3(s.CH + 3 * c);
#pragma unroll
                for (int e = 0; e < 6; ++e) ci[e] += s.CI[6 * c + e];
            }
            st3(s.FF + 3 * i, f); st3(s.NN + 3 * i, n); st3(s.CH + 3 * i, ch); s.CM[i] = cm;
#pragma unroll
            for (int e = 0; e < 6; ++e) s.CI[6 * i + e] = ci[e];
        }
        wave_sync();
    }
    if (lane < nq) {
        const int i = lane;
        const V3 z = ld3(s.Z + 3 * i), p = ld3(s.P + 3 * i), ch = ld3(s.CH + 3 * i);
        s.RHS[i] = -dot(z, ld3(s.NN + 3 * i)) - t.damping[i] * s.SQD[i];
        // momentum of the composite body below joint i for unit joint rate:
        // angular (about the world origin) K = I z + h x vO, linear Lf = m vO - h x z
        const V3 vo = cross(p, z);
        st3(s.VO + 3 * i, vo);
        st3(s.FK + 3 * i, symmul(s.CI + 6 * i, z) + cross(ch, vo));
        st3(s.FL + 3 * i, vo * s.CM[i] - cross(ch, z));
    }
    wave_sync();
    // ---- mass matrix: M[a][b] = z_b . K_a + vO_b . Lf_a for b on the path to a ----
    for (int e = lane; e < nq * nq; e += 64) {
        const int i = e / nq, j = e - i * nq;
        const int a = i > j ? i : j, b = i > j ? j : i;
        float val = 0.0f;
        if ((t.anc_mask[a] >> b) & 1) {
            val = dot(ld3(s.Z + 3 * b), ld3(s.FK + 3 * a)) + dot(ld3(s.VO + 3 * b), ld3(s.FL + 3 * a));
            if (a == b) val += t.armature[a];
        }
        s.M[e] = val;
    }
    wave_sync();
    // ---- Cholesky M = L L^T in place (lower triangle), wave-parallel.  The
    //      trailing update of column c touches the pairs (i >= j > c): one
    //      contiguous range of the pair table, 64 pairs per pass.  1/L_cc is
    //      kept in RHS-adjacent scratch (SQ is free by now) for the solves. ----
    const int n_pairs = t.pair_start[nq];
    for (int c = 0; c < nq; ++c) {
        const float inv = __builtin_amdgcn_rsqf(s.M[c * nq + c]);
        wave_sync();
        if (lane > c && lane < nq) s.M[lane * nq + c] *= inv;
        if (lane == c) { s.M[c * nq + c] = s.M[c * nq + c] * inv; s.SQ[c] = inv; }   // sqrt(d) = d * rsqrt(d)
        wave_sync();
        for (int idx = t.pair_start[c + 1] + lane; idx < n_pairs; idx += 64) {
            const int ij = t.pair_ij[idx], i = ij >> 8, j = ij & 255;
            s.M[i * nq + j] -= s.M[i * nq + c] * s.M[j * nq + c];
        }
        wave_sync();
    }
    // ---- forward and backward substitution on RHS (lane i owns row i) ----
    float bi = lane < nq ? s.RHS[lane] : 0.0f;
    for (int c = 0; c < nq; ++c) {
        const float yc = __shfl(bi, c, 64) * s.SQ[c];
        if (lane == c) bi = yc;
        else if (lane > c && lane < nq) bi -= s.M[lane * nq + c] * yc;
    }
    for (int c = nq - 1; c >= 0; --c) {
        const float xc = __shfl(bi, c, 64) * s.SQ[c];
        if (lane == c) bi = xc;
        else if (lane < c) bi -= s.M[c * nq + lane] * xc;
    }
    wave_sync();
    return bi;
}

// one env step of the own joint (lane < n_q): n_substeps integrator substeps with
// the set-points held, velocity saturation and joint limits; returns false in
// the lanes whose joint hit a limit
template <int INTEG>
__device__ __forceinline__ bool tree_integrate(const TreeDev &t, const Lds &s, int lane, float &qj, float &vj, float spk) {
    const bool joint = lane < t.n_q;
    const float vmax = joint ? t.qdmax[lane] : 0.0f, lo = joint ? t.qlo[lane] : 0.0f, hi = joint ? t.qhi[lane] : 0.0f;
    const float h = t.h;
    bool ok = true;
    auto sat = [&](float v) { return fminf(fmaxf(v, -vmax), vmax); };
    for (int sub = 0; sub < t.nsub; ++sub) {
        if (INTEG == 0) {
            const float a = tree_accel(t, s, lane, qj, vj, spk);
            vj = sat(vj + h * a);
            qj = qj + h * vj;
        } else {
            const float hh = 0.5f * h;
            const float k1q = sat(vj);
            const float k1v = tree_accel(t, s, lane, qj, k1q, spk);
            const float k2q = sat(vj + hh * k1v);
            const float k2v = tree_accel(t, s, lane, qj + hh * k1q, k2q, spk);
            const float k3q = sat(vj + hh * k2v);
            const float k3v = tree_accel(t, s, lane, qj + hh * k2q, k3q, spk);
            const float k4q = sat(vj + h * k3v);
            const float k4v = tree_accel(t, s, lane, qj + h * k3q, k4q, spk);
            const float h6 = h * (1.0f / 6.0f);
            qj = qj + h6 * (k1q + 2.0f * k2q + 2.0f * k3q + k4q);
            vj = vj + h6 * (k1v + 2.0f * k2v + 2.0f * k3v + k4v);
        }
        // velocity saturation + joint limits
        float v = sat(vj);
        const bool over = qj > hi, under = qj < lo;
        if (over) { qj = hi; v = fminf(v, 0.0f); }
        if (under) { qj = lo; v = fmaxf(v, 0.0f); }
        vj = v;
        ok = ok && !(joint && (over || under));
    }
    return ok;
}

template <int INTEG>
__global__ void __launch_bounds__(64 * TREE_WAVES)
tree_step_wave_per_env(const TreeDev tg, float *__restrict__ q, float *__restrict__ qd, uint32_t *__restrict__ feas,
                       const float *__restrict__ act, float act_scale, long n) {
    extern __shared__ float4 lds_raw4[];
    float *lds_raw = reinterpret_cast<float *>(lds_raw4);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const TreeDev t = stage_tables(tg, lds_raw, threadIdx.x, 64 * TREE_WAVES);
    const Lds s(lds_raw + 4 * tg.n_vec4 + wave * tree_lds_floats_dev(t.n_q, t.n_vp), t.n_q, t.n_vp);
    long e = long(blockIdx.x) * TREE_WAVES + wave;
    const bool live = e < n;          // a dead wave shadows the last env and stores nothing
    if (!live) e = n - 1;
    const bool joint = lane < t.n_q;
    float qj = joint ? q[long(lane) * n + e] : 0.0f;
    float vj = joint ? qd[long(lane) * n + e] : 0.0f;
    const float spk = lane < t.n_t ? act[e * t.n_t + lane] * act_scale : 0.0f;
    const bool all_ok = __all(tree_integrate<INTEG>(t, s, lane, qj, vj, spk));
    if (live && joint) { q[long(lane) * n + e] = qj; qd[long(lane) * n + e] = vj; }
    if (live && lane == 0) feas[e] = all_ok ? 1u : 0u;
}

// RoboyEnv.step fused around the tree step (the env layer of roboy_sim.hip's
// msj_env_step_kernel for joint-tree robots; same semantics):
// lane k rescales action k, lanes j hold joint j and its goal, the distances are
// wave sums, every lane evaluates the (wave-uniform) reward / done, lane j draws
// its own goal component on done.
template <int INTEG>
__global__ void __launch_bounds__(64 * TREE_WAVES)
tree_env_step_wave_per_env(const TreeDev tg, const rbe::EnvParams ep, const rbe::GoalBox box,
                           float *__restrict__ q, float *__restrict__ qd, uint32_t *__restrict__ feas,
                           float *__restrict__ goal, uint32_t *__restrict__ step_num, float *__restrict__ ep_ret,
                           uint32_t *__restrict__ goal_count, const float *__restrict__ act,
                           float *__restrict__ obs, float *__restrict__ reward, uint32_t *__restrict__ done,
                           double *__restrict__ ep_sum, uint32_t *__restrict__ ep_cnt, uint32_t *__restrict__ infeas_n,
                           long n, uint64_t seed, uint64_t env0) {
    extern __shared__ float4 lds_raw4[];
    float *lds_raw = reinterpret_cast<float *>(lds_raw4);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const TreeDev t = stage_tables(tg, lds_raw, threadIdx.x, 64 * TREE_WAVES);
    const int nq = t.n_q;
    const Lds s(lds_raw + 4 * tg.n_vec4 + wave * tree_lds_floats_dev(nq, t.n_vp), nq, t.n_vp);
    const long e = long(blockIdx.x) * TREE_WAVES + wave;
    if (e >= n) return;               // whole wave leaves together (after the staging barrier)
    const bool joint = lane < nq;
    float qj = joint ? q[long(lane) * n + e] : 0.0f;
    float vj = joint ? qd[long(lane) * n + e] : 0.0f;
    float gj = joint ? goal[long(lane) * n + e] : 0.0f;
    float spk = 0.0f;
    if (lane < t.n_t) {
        // clamp to the action box, then slope * (x - in_high) + out_high with two roundings (roboy_env.py:157-158)
        const float x = fminf(fmaxf(act[e * t.n_t + lane], -1.0f), 1.0f);
        spk = rbe::mul_then_add(ep.slope, x - 1.0f, ep.act_hi);
    }
    const bool ok = __all(tree_integrate<INTEG>(t, s, lane, qj, vj, spk));
    uint32_t sn = step_num[e] + 1u;
    const float dq = joint ? qj - gj : 0.0f;
    const float dq2 = rbe::wave_sum(dq * dq), dv2 = rbe::wave_sum(joint ? vj * vj : 0.0f);
    bool reached;
    const float r = rbe::env_reward(ep, dq2, dv2, ok, reached);
    const bool dn = reached || (sn > uint32_t(ep.max_len));
    float oq = qj, ov = vj, og = gj, ret = ep_ret[e] + r;
    uint32_t fz = ok ? 1u : 0u;
    if (dn) {   // wave-uniform
        const uint64_t gid = env0 + uint64_t(e);
        uint32_t draw = goal_count[e];
        auto draw_goal = [&](uint32_t d) {
            const rb::Philox4 rnd = rb::philox_draw(seed, gid, d, rb::STREAM_GOALS, uint32_t(lane >> 2));
            return joint ? rbe::goal_value(box.lo[lane], box.hi[lane], rnd.v[lane & 3]) : 0.0f;
        };
        gj = draw_goal(draw++);                       // RoboyEnv.step: _set_new_goal (:67-68)
        if (lane == 0) {
            ep_sum[e] += double(ret); ep_sum[n + e] += double(ret) * double(ret);
            ep_cnt[e] += 1u; ep_cnt[n + e] += sn - 1u; ep_cnt[2 * n + e] += reached ? 1u : 0u;
        }
        if (ep.auto_reset) {                          // VecEnv worker: env.reset() (:82-87)
            gj = draw_goal(draw++);
            qj = 0.0f; vj = 0.0f; oq = 0.0f; ov = 0.0f; og = gj;
            sn = 1u; fz = 1u;
        }
        ret = 0.0f;
        if (lane == 0) goal_count[e] = draw;
        if (joint) goal[long(lane) * n + e] = gj;
    }
    if (joint) {
        q[long(lane) * n + e] = qj; qd[long(lane) * n + e] = vj;
        float *orow = obs + e * (3 * nq);
        orow[lane] = oq; orow[nq + lane] = ov; orow[2 * nq + lane] = og;
    }
    if (lane == 0) {
        feas[e] = fz; step_num[e] = sn; ep_ret[e] = ret; reward[e] = r; done[e] = dn ? 1u : 0u;
        if (!ok) infeas_n[e] += 1u;
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
            q[j * n + i] = 0.0f; qd[j * n + i] = 0.0f; goal[j * n + i] = g;
            if (obs) { obs[i * 3 * n_q + j] = 0.0f; obs[i * 3 * n_q + n_q + j] = 0.0f; obs[i * 3 * n_q + 2 * n_q + j] = g; }
        }
    }
    feas[i] = 1u; step_num[i] = 1u; ep_ret[i] = 0.0f;
}

}  // namespace rbt
