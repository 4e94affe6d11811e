// tree_build.hpp - host side of the joint-tree kernels (tree_aba.hpp): constants, the kernarg struct and
// tree_build(), which flattens a robot description (rb_robot_desc) into the table buffer the kernels stage
// into LDS - tree levels and octet slots (chains of links keep their slot), per-phase level records, tendon
// crossings with their folded constant lengths, per-link gather lists, LDS layout of an env's working set.
// Plain C++ (no HIP): compiled by hipcc into the library and by g++ for tests/test_tree_tables.py, which
// checks the tables against the oracle's geometry.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/roboy_sim.h"

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#else
struct float4 { float x, y, z, w; };
#endif

namespace rbt {

constexpr int MAXQ = 32;    // joints per robot
constexpr int MAXT = 64;    // tendons per robot
constexpr int MAXVP = 1024; // via-points
#ifndef RB_TREE_E
#define RB_TREE_E 2
#endif
#ifndef RB_TREE_MIN_WAVES
#define RB_TREE_MIN_WAVES 4   // waves per SIMD the register allocation must allow (16 per CU: 8 192 upper-body envs in one go)
#endif
#ifndef RB_TREE_SKIP
#define RB_TREE_SKIP 0      // timing-only builds: bit k set = phase P(k+1) left out (results wrong by construction)
#endif
constexpr int TREE_E = RB_TREE_E;   // envs per wave
constexpr int LS = 37;      // floats per link in an env's working set (layout below); odd: lanes that hold
                            // different links of an env hit different LDS banks (strides of 64 / 48 floats
                            // cost 20- / 10-way conflicts in the per-link phases)
constexpr int TENDON_REC = 8, CROSS_REC = 8, REC1 = 12, REC5 = 24, XSLOT = 42, JOINT_REC = 8;
constexpr int ROT = 9;      // floats of a joint's rotation matrix (by columns) in the env block during P1, behind SQ

// Link block (LS floats), by phase:
//   [ 0.. 8]  R (row-major)     [ 9..11]  p      [12..14]  w      [15..17]  vO       P1 -> P5
//             then  U (6) at [0..5], 1/D, u, qdd at [6..8], a (6) at [9..14]        P5 / P6 -> P7
//   [18..20]  z    [21..23]  sl = p x z    [24..26]  c angular    [27..29]  c linear  P1 -> P6
//   [30..35]  pT: tendon wrenches on the link (added to the bias force)              P4 -> P5
// Rows / components: 0-2 angular, 3-5 linear.
constexpr int O_RP = 0, O_V = 12, O_U = 0, O_D = 6, O_A = 9, O_S = 18, O_C = 24, O_PT = 30;
// level-record flags
constexpr int F_INH = 1;        // the parent sits in the same octet one level up: its data arrive in registers
constexpr int F_ASTORE = 2;     // some child sits in another octet: the acceleration goes through LDS as well
constexpr int F_REGCHILD = 1;   // (P5) one child sits in the same octet one level down: its I^a / p^a arrive in registers
constexpr int F_XWRITE = 2;     // (P5) the parent sits in another octet: I^a / p^a go to the link's exchange slot
constexpr int F_MASSLESS = 4;   // (P5) no mass, no inertia (the x / y joints of a ball joint): nothing of its own to add

struct TreeDev {
    int n_q, n_t, n_cr, n_levels, nsub, n_x;   // n_x: exchange slots (links whose parent sits in another octet)
    int single_pass;                 // a tree level fits one pass of the wave (E * lw octets <= 8): the SP kernels
    int lw_shift, q_shift;           // log2 of the octets per env in the level passes / of the lane slots per env in the joint passes
    int ES, o_W, o_SQD, o_SPU, zoff; // env stride (floats) and offsets inside an env's block (SQ and the exchange slots alias W)
    int o_lc_start, o_lc_list, o_lc_link, o_t_cr_start, o_rec1, o_rec5, o_ext_list, o_tendon, o_cross, o_joint;   // word offsets into the table buffer
    float h, g[3], kps, pe_k2s, inv_pe_den, fv_c1l, fv_c2l, fv_c2s, fv_k;
    const float4 *g_words;           // all tables as one device buffer of 32-bit words, staged to LDS per workgroup
    int n_vec4;
};

struct TreeHost {
    std::vector<uint32_t> words;
    TreeDev dev;
    size_t ws_floats_per_wave = 0;   // E * ES
    size_t table_floats = 0;         // padded to 16 bytes
};

inline uint32_t f2w(double v) { const float f = float(v); uint32_t w; std::memcpy(&w, &f, 4); return w; }

// Flatten the description; rest lengths and every derived constant in fp64.
inline int tree_build(const rb_robot_desc *d, double step_size, int nsub, TreeHost &out, std::string &err) {
    const int nq = d->n_q, nt = d->n_t, nvp = d->n_vp;
    if (nq < 1 || nq > MAXQ) { err = "generic-tree kernel supports 1..32 joints"; return RB_EUNSUPPORTED; }
    if (nt < 1 || nt > MAXT) { err = "generic-tree kernel supports 1..64 tendons"; return RB_EUNSUPPORTED; }
    if (nvp > MAXVP) { err = "too many via-points"; return RB_EUNSUPPORTED; }
    std::vector<int> level(nq), parent(d->parent, d->parent + nq);
    int nlev = 0;
    for (int i = 0; i < nq; ++i) {
        if (parent[i] < -1 || parent[i] >= i) { err = "parent must be -1 or an earlier joint"; return RB_EINVAL; }
        level[i] = parent[i] < 0 ? 0 : level[parent[i]] + 1;
        nlev = level[i] + 1 > nlev ? level[i] + 1 : nlev;
    }
    std::vector<std::vector<int>> by_level(nlev), children(nq);
    int max_width = 1;
    for (int i = 0; i < nq; ++i) {
        by_level[level[i]].push_back(i);
        if (parent[i] >= 0) children[parent[i]].push_back(i);
    }
    for (int L = 0; L < nlev; ++L) max_width = int(by_level[L].size()) > max_width ? int(by_level[L].size()) : max_width;
    int lw = 1, lw_shift = 0;
    while (lw < max_width) { lw <<= 1; ++lw_shift; }
    // Octet slots of the level passes.  A link inherits its parent's slot when it can (first child
    // served), so that a chain of links stays in one octet and hands its data on in registers; that
    // needs the whole level to fit one pass of the wave (E * lw octets of 8 lanes <= 64 lanes).
    const bool chain_ok = TREE_E * lw * 8 <= 64;
    std::vector<int> slot(nq, -1), inh(nq, 0), reg_child(nq, -1);
    for (int L = 0; L < nlev; ++L) {
        std::vector<char> used(lw, 0);
        if (chain_ok)
            for (int i : by_level[L]) {
                const int par = parent[i];
                if (par >= 0 && reg_child[par] < 0 && !used[slot[par]]) { slot[i] = slot[par]; used[slot[i]] = 1; inh[i] = 1; reg_child[par] = i; }
            }
        for (int i : by_level[L]) {
            if (slot[i] >= 0) continue;
            int x = 0;
            while (used[x]) ++x;
            slot[i] = x; used[x] = 1;
        }
    }
    // exchange slots: links that hand I^a / p^a to a parent in another octet
    std::vector<int> xslot(nq, -1);
    int n_x = 0;
    for (int i = 0; i < nq; ++i) if (parent[i] >= 0 && !inh[i]) xslot[i] = n_x++;
    // zero pose => every link frame is a pure translation: rest lengths, constant segments
    std::vector<double> org(3 * nq);
    for (int i = 0; i < nq; ++i)
        for (int a = 0; a < 3; ++a) org[3 * i + a] = (parent[i] < 0 ? 0.0 : org[3 * parent[i] + a]) + d->origin[3 * i + a];
    const double log2e = 1.4426950408889634;
    const double sc = std::sqrt(log2e) / d->fl_width;     // strain scale: f_L = exp2(-(sc e)^2), as in msj_build.hpp
    struct Cross { int la, lb; double ra[3], rb[3]; };
    std::vector<Cross> cross;
    std::vector<int> t_cr_start(nt + 1, 0);
    std::vector<double> t_rec(size_t(nt) * TENDON_REC, 0.0);
    for (int k = 0; k < nt; ++k) {
        const int v0 = d->vp_offset[k], v1 = d->vp_offset[k + 1];
        if (v1 - v0 < 2) { err = "tendon with fewer than two via-points"; return RB_EINVAL; }
        double l0 = 0.0, lconst = 0.0;
        t_cr_start[k] = int(cross.size());
        for (int v = v0; v + 1 < v1; ++v) {
            const int la = d->vp_link[v], lb = d->vp_link[v + 1];
            if (la < -1 || la >= nq || lb < -1 || lb >= nq) { err = "via-point on an unknown link"; return RB_EINVAL; }
            double s = 0.0, sl = 0.0;
            for (int a = 0; a < 3; ++a) {
                const double xa = (la < 0 ? 0.0 : org[3 * la + a]) + d->vp_pos[3 * v + a];
                const double xb = (lb < 0 ? 0.0 : org[3 * lb + a]) + d->vp_pos[3 * (v + 1) + a];
                s += (xb - xa) * (xb - xa);
                const double dl = d->vp_pos[3 * (v + 1) + a] - d->vp_pos[3 * v + a];
                sl += dl * dl;
            }
            if (s < 1e-12) { err = "degenerate tendon segment"; return RB_EINVAL; }
            l0 += std::sqrt(s);
            if (la == lb) {
                lconst += std::sqrt(sl);        // both ends move with the same link: constant length, no net wrench
            } else {
                Cross c;
                c.la = la; c.lb = lb;
                for (int a = 0; a < 3; ++a) { c.ra[a] = d->vp_pos[3 * v + a]; c.rb[a] = d->vp_pos[3 * (v + 1) + a]; }
                cross.push_back(c);
            }
        }
        double *r = &t_rec[size_t(k) * TENDON_REC];
        r[0] = sc / l0;                              // il0s
        r[1] = sc * (lconst / l0 - 1.0);             // elcs
        r[2] = d->kp * d->setpoint_scale / l0;       // ksg
        r[3] = d->f_max[k];
        r[4] = 1.0 / (d->v_max * l0);                // inv_vl0
    }
    t_cr_start[nt] = int(cross.size());
    const int ncr = int(cross.size());
    // P4's lists: for every link, the crossings incident to it with the sign of their wrench in p^A
    // (fext_la += W, fext_lb -= W, p^A -= fext): entry = W offset (6 * crossing) << 1 | (1 if the link is lb,
    // i.e. +W).  Links sorted by falling list length (lanes of one pass then loop about equally long),
    // every list padded to a multiple of 4 with entries that point at the env block's zero slot, so the
    // gather loop runs 4 independent loads per trip.  Links without tendons have empty lists (pT = 0).
    const int zoff = 6 * ncr > nq ? 6 * ncr : nq;      // zero slot: behind W and behind SQ, which aliases W's start
    std::vector<std::vector<int>> inc(nq);
    for (int c = 0; c < ncr; ++c) {
        if (cross[c].la >= 0) inc[cross[c].la].push_back((6 * c) << 1);
        if (cross[c].lb >= 0) inc[cross[c].lb].push_back(((6 * c) << 1) | 1);
    }
    std::vector<int> lc_link(nq);
    for (int i = 0; i < nq; ++i) lc_link[i] = i;
    std::stable_sort(lc_link.begin(), lc_link.end(), [&](int a, int b) { return inc[a].size() > inc[b].size(); });
    std::vector<int> lc_start, lc_list;
    for (int i : lc_link) {
        lc_start.push_back(int(lc_list.size()));
        for (int en : inc[i]) lc_list.push_back(en);
        while (lc_list.size() % 4) lc_list.push_back(zoff << 1);
    }
    lc_start.push_back(int(lc_list.size()));
    if (lc_list.empty()) lc_list.assign(4, zoff << 1);

    std::vector<uint32_t> &w = out.words;
    w.clear();
    auto pad4 = [&]() { while (w.size() % 4) w.push_back(0u); };
    auto push_i = [&](const std::vector<int> &v) { pad4(); const int off = int(w.size()); for (int x : v) w.push_back(uint32_t(x)); return off; };
    TreeDev &t = out.dev;
    t.o_lc_start = push_i(lc_start); t.o_lc_list = push_i(lc_list); t.o_lc_link = push_i(lc_link);
    t.o_t_cr_start = push_i(t_cr_start);
    // exchange slots of the children that sit in another octet, per link
    std::vector<int> ext_start(nq + 1, 0), ext_list;
    for (int i = 0; i < nq; ++i) {
        ext_start[i] = int(ext_list.size());
        for (int c : children[i]) if (!inh[c]) ext_list.push_back(xslot[c]);
    }
    ext_start[nq] = int(ext_list.size());
    if (ext_list.empty()) ext_list.push_back(0);
    t.o_ext_list = push_i(ext_list);
    // level records, one per (level, octet slot x < lw); empty slots have i = -1
    std::vector<int> at(size_t(nlev) * lw, -1);
    for (int i = 0; i < nq; ++i) at[size_t(level[i]) * lw + slot[i]] = i;
    pad4(); t.o_rec1 = int(w.size());
    for (int q = 0; q < nlev * lw; ++q) {          // P1 / P6: i, parent, flags, 0, axis 3, origin 3, 0, 0
        const int i = at[q];
        const bool used = i >= 0;
        bool astore = false;
        if (used) for (int c : children[i]) astore = astore || !inh[c];
        w.push_back(uint32_t(i)); w.push_back(uint32_t(used ? parent[i] : -1));
        w.push_back(uint32_t(used ? (inh[i] ? F_INH : 0) | (astore ? F_ASTORE : 0) : 0)); w.push_back(0u);
        for (int a = 0; a < 3; ++a) w.push_back(f2w(used ? d->axis[3 * i + a] : 0.0));
        for (int a = 0; a < 3; ++a) w.push_back(f2w(used ? d->origin[3 * i + a] : 0.0));
        w.push_back(0u); w.push_back(0u);
    }
    pad4(); t.o_rec5 = int(w.size());
    for (int q = 0; q < nlev * lw; ++q) {          // P5: i, parent, flags, n_ext, ext 0..3, ext_start, xslot, 0, 0, mass, com 3, inertia 6, armature, damping
        const int i = at[q];
        const bool used = i >= 0;
        const int es = used ? ext_start[i] : 0, ne = used ? ext_start[i + 1] - es : 0;
        w.push_back(uint32_t(i)); w.push_back(uint32_t(used ? parent[i] : -1));
        const bool massless = used && d->mass[i] == 0.0 && d->inertia[6 * i] == 0.0 && d->inertia[6 * i + 1] == 0.0 && d->inertia[6 * i + 2] == 0.0;
        w.push_back(uint32_t(used ? (reg_child[i] >= 0 ? F_REGCHILD : 0) | (xslot[i] >= 0 ? F_XWRITE : 0) | (massless ? F_MASSLESS : 0) : 0));
        w.push_back(uint32_t(ne));
        for (int k = 0; k < 4; ++k) w.push_back(uint32_t(k < ne ? ext_list[es + k] : 0));
        w.push_back(uint32_t(es)); w.push_back(uint32_t(used && xslot[i] >= 0 ? xslot[i] : 0)); w.push_back(0u); w.push_back(0u);
        w.push_back(f2w(used ? d->mass[i] : 0.0));
        for (int a = 0; a < 3; ++a) w.push_back(f2w(used ? d->com[3 * i + a] : 0.0));
        for (int a = 0; a < 6; ++a) w.push_back(f2w(used ? d->inertia[6 * i + a] : 0.0));
        w.push_back(f2w(used ? d->armature[i] : 1.0)); w.push_back(f2w(used ? d->damping[i] : 0.0));
    }
    pad4(); t.o_joint = int(w.size());
    for (int i = 0; i < nq; ++i) {                  // per joint: qlo, qhi, qdmax, 0, axis 3, 0
        w.push_back(f2w(d->q_lo[i])); w.push_back(f2w(d->q_hi[i])); w.push_back(f2w(d->qd_max[i])); w.push_back(0u);
        for (int a = 0; a < 3; ++a) w.push_back(f2w(d->axis[3 * i + a]));
        w.push_back(0u);
    }
    pad4(); t.o_tendon = int(w.size());
    for (double x : t_rec) w.push_back(f2w(x));
    pad4(); t.o_cross = int(w.size());
    for (const Cross &c : cross) {
        w.push_back(uint32_t(c.la)); w.push_back(uint32_t(c.lb));
        for (int a = 0; a < 3; ++a) w.push_back(f2w(c.ra[a]));
        for (int a = 0; a < 3; ++a) w.push_back(f2w(c.rb[a]));
    }
    pad4();
    out.table_floats = w.size();

    t.n_q = nq; t.n_t = nt; t.n_cr = ncr; t.n_levels = nlev; t.nsub = nsub; t.n_x = n_x; t.h = float(step_size / nsub);
    t.lw_shift = lw_shift;
    t.single_pass = chain_ok ? 1 : 0;
    int qw = 1; t.q_shift = 0;
    while (qw < nq) { qw <<= 1; ++t.q_shift; }
    // env block: links | W (6 per crossing, then a zero slot of 6; SQ and the joints' rotation matrices alias the
    // start during P1, the exchange slots alias it during P5) | SQD | SPU
    int wsz = zoff + 6;
    if (XSLOT * n_x > wsz) wsz = XSLOT * n_x;
    if ((1 + ROT) * nq > wsz) wsz = (1 + ROT) * nq;   // P1: SQ (nq) and behind it the joints' rotation matrices
    t.zoff = zoff;
    t.o_W = nq * LS;
    t.o_SQD = t.o_W + wsz;
    t.o_SPU = t.o_SQD + nq;
    t.ES = t.o_SPU + nt;
    if (t.ES % 2 == 0) ++t.ES;                      // odd: the envs of a wave start in different banks
    for (int a = 0; a < 3; ++a) t.g[a] = float(d->gravity[a]);
    t.kps = float(d->kp / sc);
    t.pe_k2s = float(log2e * d->kpe / (d->e0 * sc));
    t.inv_pe_den = float(1.0 / (std::exp(d->kpe) - 1.0));
    const double c2l = (1.0 + 1.0 / d->fv_a) / (d->fv_n - 1.0);
    t.fv_c2s = float(-1.0 / d->fv_a); t.fv_k = float(1.0 + 1.0 / d->fv_a);
    t.fv_c1l = float(d->fv_n * c2l); t.fv_c2l = float(c2l);
    t.g_words = nullptr; t.n_vec4 = int(w.size() / 4);
    out.ws_floats_per_wave = size_t(TREE_E) * t.ES;
    return RB_OK;
}

// dynamic LDS of a workgroup of `waves` waves
inline size_t tree_lds_bytes(const TreeHost &h, int waves) { return 4 * (h.table_floats + size_t(waves) * h.ws_floats_per_wave); }

// waves per workgroup (1..8) that keeps the most waves resident on a CU (160 KiB of LDS, one table copy per workgroup)
inline int tree_pick_waves(const TreeHost &h) {
    int best = 1, best_res = 0;
    for (int wv = 1; wv <= 8; ++wv) {
        const size_t bytes = tree_lds_bytes(h, wv);
        if (bytes > 160 * 1024) break;
        int wgs = int((160 * 1024) / bytes);
        int res = wgs * wv;
        if (res > 32) res = 32;
        if (res > best_res || (res == best_res && wv < best)) { best_res = res; best = wv; }
    }
    return best;
}

}  // namespace rbt
