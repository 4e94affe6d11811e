// tree_lane_gen.hpp - source generator of the env-per-lane joint-tree kernels (tree_lane.hpp).
//
// The octet kernels of tree_aba.hpp spend ~9 % of their lane slots on arithmetic (profiles/r2_a: 1 516 vector
// wave-instructions per upper-body env against ~8 700 lane-instructions of work): level records, table lookups,
// DPP traffic and idle lanes are the rest.  This generator removes all of it for ONE robot: it walks the robot
// description on the host and writes the articulated-body algorithm (world coordinates about the world origin,
// the formulation of tree_aba.hpp / tests/proto/aba_world.py) as straight-line single-assignment C++ for ONE env,
// which a lane then executes for its env:
//   * the topology is gone - every link's step is its own code, parent data are plain values;
//   * every robot constant is a hex-float literal, and the generator folds them while it writes (a product
//     with a literal 0 or 1 is never emitted): axis-aligned joints, zero joint origins, massless links and
//     diagonal inertias cost nothing.  The upper body's acceleration is ~6 800 vector instructions per wave
//     = 106 per env, against 1 516 per env in the octet form;
//   * tendons that cross between the same pair of links share the pair's relative velocity, and their wrenches
//     are summed per pair before they are applied to the two links.  With m = x_a x u (= x_b x u, as u is
//     parallel to x_b - x_a) the length rate is u.(vO_b - vO_a) + m.(w_b - w_a) and the unit wrench (m ; u);
//   * values that have to survive from the forward-kinematics sweep to the backward and forward passes but are
//     not used in between (the velocity-product accelerations c) go to lane-private LDS columns (RBL_LDS(slot)),
//     everything else is left to the register allocator (512 VGPR + AGPR at one wave per SIMD).
// The text is compiled three ways: by hipcc into the library for the committed upper body (tree_lane_baked.hpp,
// written by tools/gen_tree_lane_baked.py), by hiprtc at run time for any other robot (tree_lane_jit.hpp), and by
// g++ for the CPU tests (tests/test_tree_lane_gen.py), which check it against the fp64 oracle on random robots.
// Plain host C++ (no HIP).
#pragma once
#include <array>
#include <cctype>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <map>
#include <string>
#include <utility>
#include <vector>

#include "../../include/roboy_sim.h"

namespace rblg {

struct Val {
    bool k = true;       // a compile-time constant (c) / a named value (name, possibly negated)
    double c = 0.0;
    std::string name;
    bool neg = false;
};

using V3 = std::array<Val, 3>;
using M3 = std::array<Val, 9>;   // row-major

struct Generated {
    std::string text;          // the header: macros, tables and rbl_accel
    int n_q = 0, n_t = 0;
    int lds_slots = 0;         // lane-private LDS slots rbl_accel uses
    int n_stmt = 0;            // statements emitted (a proxy of the instruction count)
    int max_live = 0;          // most temporaries alive at once in the order the text is written (the sched barriers keep the
                               // compiler close to it): what has to fit the 512 registers of a SIMD's only wave, beside the
                               // kernel's own ~60 (inputs, addresses).  Robots beyond ~400 spill to scratch - slow to build
                               // (minutes) and slow to run (every reload a full memory latency) - and keep the octet kernels.
    int flops = 0;             // floating-point operations of one acceleration as written (one per arithmetic statement: the
                               // folded products and sums are not counted - they are not executed)
    uint64_t hash = 0;         // FNV-1a of the text
};

class Gen {
 public:
    // one statement of the generated function: the temporary it defines ("" for a store, which is always kept) and its text
    struct Stmt { std::string target, text; };
    std::vector<Stmt> stmts;
    int n_tmp = 0, n_lds = 0;

    static Val K(double c) { Val v; v.k = true; v.c = c; return v; }
    static Val named(const std::string &n) { Val v; v.k = false; v.name = n; return v; }
    static std::string lit(double c) {
        char buf[64];
        const float f = float(c);
        if (f == 0.0f) return "0.0f";
        std::snprintf(buf, sizeof buf, "%af", double(f < 0 ? -f : f));
        return std::string(f < 0 ? "-" : "") + buf;
    }
    static bool is0(const Val &v) { return v.k && float(v.c) == 0.0f; }
    static bool is1(const Val &v) { return v.k && v.c == 1.0; }
    static bool ism1(const Val &v) { return v.k && v.c == -1.0; }
    // operand text
    static std::string S(const Val &v) {
        if (v.k) { const std::string l = lit(v.c); return l[0] == '-' ? "(" + l + ")" : l; }
        return v.neg ? "(-" + v.name + ")" : v.name;
    }
    Val emit(const std::string &expr) {
        const std::string n = "t" + std::to_string(n_tmp++);
        stmts.push_back({n, "    const float " + n + " = " + expr + ";\n"});
        return named(n);
    }
    void store(const std::string &lhs, const Val &v) { stmts.push_back({"", "    " + lhs + " = " + S(v) + ";\n"}); }
    void comment(const std::string &s) { stmts.push_back({"//", "    // " + s + "\n"}); }
    // the instruction scheduler must not move work across this point: the text is written in an order that keeps
    // few values alive (a link's frame dies as soon as its children and tendons are done), and the kernel has no
    // registers to spare for a scheduler that interleaves the groups for latency (one wave per SIMD: 512 registers)
    void barrier() { stmts.push_back({"//", "    RBL_SCHED_BARRIER;\n"}); }
    // The text of the function body without the statements nothing depends on (a parent's accumulated inertia that
    // only feeds a root's unused I^a, products folded away further down, ...); n_stmt = statements kept.
    std::string body(int &n_stmt, int &flops, int &max_live) const {
        std::vector<char> keep(stmts.size(), 0);
        std::map<std::string, char> live;
        for (size_t k = stmts.size(); k-- > 0;) {
            const Stmt &st = stmts[k];
            if (st.target == "//") { keep[k] = 1; continue; }
            if (!st.target.empty() && !live.count(st.target)) continue;
            keep[k] = 1;
            // every temporary named in the text (the defined one included: harmless) is needed
            const std::string &t = st.text;
            for (size_t i = 0; i < t.size(); ++i)
                if (t[i] == 't' && i + 1 < t.size() && t[i + 1] >= '0' && t[i + 1] <= '9' && (i == 0 || !(isalnum((unsigned char)t[i - 1]) || t[i - 1] == '_'))) {
                    size_t j = i + 1;
                    while (j < t.size() && t[j] >= '0' && t[j] <= '9') ++j;
                    live[t.substr(i, j - i)] = 1;
                    i = j - 1;
                }
        }
        // live ranges in emission order: a temporary lives from its definition to its last use
        {
            std::map<std::string, size_t> last_use;
            auto names_in = [](const std::string &t, std::vector<std::string> &out) {
                for (size_t i = 0; i < t.size(); ++i)
                    if (t[i] == 't' && i + 1 < t.size() && t[i + 1] >= '0' && t[i + 1] <= '9' && (i == 0 || !(isalnum((unsigned char)t[i - 1]) || t[i - 1] == '_'))) {
                        size_t j = i + 1;
                        while (j < t.size() && t[j] >= '0' && t[j] <= '9') ++j;
                        out.push_back(t.substr(i, j - i));
                        i = j - 1;
                    }
            };
            for (size_t k = 0; k < stmts.size(); ++k) {
                if (!keep[k] || stmts[k].target == "//") continue;
                std::vector<std::string> ns;
                names_in(stmts[k].text, ns);
                for (const std::string &n : ns) last_use[n] = k;
            }
            std::vector<int> dies(stmts.size() + 1, 0);
            for (const auto &kv : last_use) ++dies[kv.second];
            int live = 0;
            max_live = 0;
            for (size_t k = 0; k < stmts.size(); ++k) {
                if (!keep[k] || stmts[k].target == "//") continue;
                if (!stmts[k].target.empty()) ++live;
                if (live > max_live) max_live = live;
                live -= dies[k];
            }
        }
        std::string out;
        n_stmt = 0; flops = 0;
        for (size_t k = 0; k < stmts.size(); ++k)
            if (keep[k]) {
                out += stmts[k].text;
                if (stmts[k].target == "//") continue;
                ++n_stmt;
                // arithmetic = a kept temporary that is not an LDS read-back
                if (!stmts[k].target.empty() && stmts[k].text.find("RBL_LDS(") == std::string::npos) ++flops;
            }
        return out;
    }
    static Val negv(Val a) {
        if (a.k) { a.c = -a.c; return a; }
        a.neg = !a.neg;
        return a;
    }
    Val add(const Val &a, const Val &b) {
        if (a.k && b.k) return K(a.c + b.c);
        if (is0(a)) return b;
        if (is0(b)) return a;
        if (a.k) return add(b, a);                       // named first
        if (b.k) {                                       // named + constant
            if (!a.neg) return emit(a.name + (b.c < 0 ? " - " + lit(-b.c) : " + " + lit(b.c)));
            return emit(lit(b.c) + " - " + a.name);
        }
        if (!a.neg && !b.neg) return emit(a.name + " + " + b.name);
        if (!a.neg && b.neg) return emit(a.name + " - " + b.name);
        if (a.neg && !b.neg) return emit(b.name + " - " + a.name);
        return negv(emit(a.name + " + " + b.name));
    }
    Val sub(const Val &a, const Val &b) { return add(a, negv(b)); }
    Val mul(const Val &a, const Val &b) {
        if (a.k && b.k) return K(a.c * b.c);
        if (is0(a) || is0(b)) return K(0.0);
        if (a.k) return mul(b, a);
        if (b.k) {
            if (is1(b)) return a;
            if (ism1(b)) return negv(a);
            Val r = emit(a.name + " * " + lit(std::fabs(b.c)));
            r.neg = a.neg != (b.c < 0);
            return r;
        }
        Val r = emit(a.name + " * " + b.name);
        r.neg = a.neg != b.neg;
        return r;
    }
    Val fma(const Val &a, const Val &b, const Val &c) { return add(mul(a, b), c); }
    // sum of products, accumulated left to right
    Val dot(const std::vector<std::pair<Val, Val>> &terms) {
        Val acc = K(0.0);
        for (const auto &t : terms) acc = fma(t.first, t.second, acc);
        return acc;
    }
    Val call1(const char *fn, const Val &a) { return emit(std::string(fn) + "(" + S(a) + ")"); }
    Val call2(const char *fn, const Val &a, const Val &b) { return emit(std::string(fn) + "(" + S(a) + ", " + S(b) + ")"); }
    Val call3(const char *fn, const Val &a, const Val &b, const Val &c) {
        return emit(std::string(fn) + "(" + S(a) + ", " + S(b) + ", " + S(c) + ")");
    }
    // a value that is used again much later: to a lane-private LDS slot and back
    int lds_store(const Val &v) {
        const int slot = n_lds++;
        store("RBL_LDS(" + std::to_string(slot) + ")", v);
        return slot;
    }
    Val lds_load(int slot) { return emit("RBL_LDS(" + std::to_string(slot) + ")"); }

    // ---- small vector algebra on symbolic values ----
    V3 vadd(const V3 &a, const V3 &b) { return {add(a[0], b[0]), add(a[1], b[1]), add(a[2], b[2])}; }
    V3 vsub(const V3 &a, const V3 &b) { return {sub(a[0], b[0]), sub(a[1], b[1]), sub(a[2], b[2])}; }
    V3 vscale(const V3 &a, const Val &s) { return {mul(a[0], s), mul(a[1], s), mul(a[2], s)}; }
    V3 vfma(const V3 &a, const Val &s, const V3 &b) { return {fma(a[0], s, b[0]), fma(a[1], s, b[1]), fma(a[2], s, b[2])}; }
    Val vdot(const V3 &a, const V3 &b) { return dot({{a[0], b[0]}, {a[1], b[1]}, {a[2], b[2]}}); }
    // (the order in which a compiler evaluates the arguments of one call is unspecified: wherever two arguments would
    // emit statements they are evaluated in statements of their own, so that the text does not depend on the compiler
    // the generator was built with - the library regenerates it at run time and compares it with the committed one)
    V3 cross(const V3 &a, const V3 &b) {
        V3 o;
        for (int k = 0; k < 3; ++k) {
            const int n1 = (k + 1) % 3, n2 = (k + 2) % 3;
            const Val l = mul(a[n1], b[n2]);
            const Val r = mul(a[n2], b[n1]);
            o[k] = sub(l, r);
        }
        return o;
    }
    V3 matvec(const M3 &m, const V3 &v) {
        return {dot({{m[0], v[0]}, {m[1], v[1]}, {m[2], v[2]}}), dot({{m[3], v[0]}, {m[4], v[1]}, {m[5], v[2]}}),
                dot({{m[6], v[0]}, {m[7], v[1]}, {m[8], v[2]}})};
    }
    M3 matmul(const M3 &a, const M3 &b) {
        M3 o;
        for (int r = 0; r < 3; ++r)
            for (int c = 0; c < 3; ++c) o[3 * r + c] = dot({{a[3 * r], b[c]}, {a[3 * r + 1], b[3 + c]}, {a[3 * r + 2], b[6 + c]}});
        return o;
    }
    static V3 KV(const double *p) { return {K(p[0]), K(p[1]), K(p[2])}; }
    static V3 zero3() { return {K(0.0), K(0.0), K(0.0)}; }
};

// symmetric 6x6 (rows / columns 0-2 angular, 3-5 linear), upper triangle stored
struct Sym6 {
    Val m[6][6];
    Val &at(int r, int c) { return r <= c ? m[r][c] : m[c][r]; }
};

inline uint64_t fnv1a(const std::string &s) {
    uint64_t h = 1469598103934665603ull;
    for (unsigned char ch : s) { h ^= ch; h *= 1099511628211ull; }
    return h;
}

// One tendon crossing between two different links (as tree_build.hpp folds them: segments inside one link have a
// constant length and no net wrench).
struct Crossing { int la, lb; double ra[3], rb[3]; };

// Write the header for robot `d`.  lds_c: keep the velocity-product accelerations in LDS between the sweeps.
inline int generate(const rb_robot_desc *d, bool lds_c, Generated &out, std::string &err) {
    const int nq = d->n_q, nt = d->n_t;
    if (nq < 1 || nq > 32) { err = "lane kernel generator supports 1..32 joints"; return RB_EUNSUPPORTED; }
    if (nt < 1 || nt > 64) { err = "lane kernel generator supports 1..64 tendons"; return RB_EUNSUPPORTED; }
    std::vector<int> parent(d->parent, d->parent + nq);
    std::vector<std::vector<int>> children(nq);
    for (int i = 0; i < nq; ++i) {
        if (parent[i] < -1 || parent[i] >= i) { err = "parent must be -1 or an earlier joint"; return RB_EINVAL; }
        if (parent[i] >= 0) children[parent[i]].push_back(i);
    }
    // ---- tendon constants in fp64 (the formulas of tree_build.hpp) ----
    std::vector<double> org(3 * nq);
    for (int i = 0; i < nq; ++i)
        for (int a = 0; a < 3; ++a) org[3 * i + a] = (parent[i] < 0 ? 0.0 : org[3 * parent[i] + a]) + d->origin[3 * i + a];
    const double log2e = 1.4426950408889634;
    const double sc = std::sqrt(log2e) / d->fl_width;
    std::vector<std::vector<Crossing>> t_cross(nt);
    std::vector<double> il0s(nt), elcs(nt), ksg(nt), inv_vl0(nt);
    for (int k = 0; k < nt; ++k) {
        const int v0 = d->vp_offset[k], v1 = d->vp_offset[k + 1];
        if (v1 - v0 < 2) { err = "tendon with fewer than two via-points"; return RB_EINVAL; }
        double l0 = 0.0, lconst = 0.0;
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
                lconst += std::sqrt(sl);
            } else {
                Crossing c;
                c.la = la; c.lb = lb;
                for (int a = 0; a < 3; ++a) { c.ra[a] = d->vp_pos[3 * v + a]; c.rb[a] = d->vp_pos[3 * (v + 1) + a]; }
                t_cross[k].push_back(c);
            }
        }
        il0s[k] = sc / l0; elcs[k] = sc * (lconst / l0 - 1.0);
        ksg[k] = d->kp * d->setpoint_scale / l0; inv_vl0[k] = 1.0 / (d->v_max * l0);
    }
    const double kps = d->kp / sc, pe_k2s = log2e * d->kpe / (d->e0 * sc), inv_pe_den = 1.0 / (std::exp(d->kpe) - 1.0);
    const double c2l = (1.0 + 1.0 / d->fv_a) / (d->fv_n - 1.0), fv_c2s = -1.0 / d->fv_a, fv_k = 1.0 + 1.0 / d->fv_a;
    const double fv_c1l = d->fv_n * c2l, fv_c2l = c2l;

    Gen g;
    auto K = [](double c) { return Gen::K(c); };
    // ---------------- sweep 1: frames, joint axes, velocities, velocity-product accelerations ----------------
    std::vector<M3> R(nq);
    std::vector<V3> p(nq), w(nq), vo(nq), z(nq), sl(nq);
    std::vector<std::array<Val, 6>> cacc(nq);
    std::vector<std::array<int, 6>> cslot(nq);
    // the link's own spatial inertia about the world origin and its bias force v x* (I v), evaluated while the frame
    // and the velocity are at hand: from here to the backward pass a massive link is carried as I_o (6), h (3) and
    // a bias force (6, which also collects the tendon wrenches) instead of R, p, w, vO and its wrenches (24)
    std::vector<Sym6> Iown(nq);
    std::vector<std::array<Val, 6>> bown(nq);
    const M3 ident = {K(1), K(0), K(0), K(0), K(1), K(0), K(0), K(0), K(1)};
    // ---- tendons: Hill force, wrench sums per link pair; a tendon is evaluated as soon as the frames of all the links
    //      it touches exist, i.e. right after the link with the largest index among them ----
    std::map<std::pair<int, int>, std::array<Val, 6>> pair_sum;      // sum of F (m ; u) over the crossings la -> lb
    std::map<std::pair<int, int>, std::pair<V3, V3>> pair_vel;       // (w_b - w_a, vO_b - vO_a)
    std::vector<std::pair<int, int>> pair_order;
    auto link_w = [&](int l) { return l < 0 ? Gen::zero3() : w[l]; };
    auto link_vo = [&](int l) { return l < 0 ? Gen::zero3() : vo[l]; };
    std::vector<int> t_last(nt, -1);                                  // the link after which tendon k can be evaluated (-1: before any)
    for (int k = 0; k < nt; ++k)
        for (const Crossing &cr : t_cross[k]) { if (cr.la > t_last[k]) t_last[k] = cr.la; if (cr.lb > t_last[k]) t_last[k] = cr.lb; }
    auto emit_tendon = [&](int k) {
        g.comment("tendon " + std::to_string(k));
        Val len = K(0.0), ldot = K(0.0);
        struct Unit { std::pair<int, int> pr; V3 m, u; };
        std::vector<Unit> units;
        for (const Crossing &cr : t_cross[k]) {
            const std::pair<int, int> pr(cr.la, cr.lb);
            if (!pair_vel.count(pr)) {
                pair_vel[pr] = {g.vsub(link_w(cr.lb), link_w(cr.la)), g.vsub(link_vo(cr.lb), link_vo(cr.la))};
                pair_sum[pr] = {K(0), K(0), K(0), K(0), K(0), K(0)};
                pair_order.push_back(pr);
            }
            const V3 xa = cr.la < 0 ? Gen::KV(cr.ra) : g.vadd(p[cr.la], g.matvec(R[cr.la], Gen::KV(cr.ra)));
            const V3 xb = cr.lb < 0 ? Gen::KV(cr.rb) : g.vadd(p[cr.lb], g.matvec(R[cr.lb], Gen::KV(cr.rb)));
            const V3 dd = g.vsub(xb, xa);
            const Val d2 = g.vdot(dd, dd), inv = g.call1("rbl_rsq", d2);
            const V3 u = g.vscale(dd, inv);
            len = g.fma(d2, inv, len);
            const V3 m = g.cross(xa, u);
            const Val ldl = g.vdot(u, pair_vel[pr].second);
            const Val lda = g.vdot(m, pair_vel[pr].first);
            ldot = g.add(ldot, g.add(ldl, lda));
            units.push_back({pr, m, u});
        }
        // Hill-type force, scaled forms as in tree_aba.hpp p2_tendon / msj_math.hpp
        const Val es = g.fma(len, K(il0s[k]), K(elcs[k]));
        const Val act = g.call3("rbl_med3", g.sub(g.mul(es, K(kps)), Gen::named("spu[" + std::to_string(k) + "]")), K(0.0), K(1.0));
        const Val fl = g.call1("rbl_exp2", Gen::negv(g.mul(es, es)));
        const Val v = g.mul(ldot, K(inv_vl0[k]));
        const Val vp = g.call2("rbl_max", v, K(0.0)), pq = g.call3("rbl_med3", g.add(v, K(1.0)), K(0.0), K(1.0));
        const Val num = g.fma(vp, K(fv_c1l), pq), den = g.fma(vp, K(fv_c2l), g.fma(pq, K(fv_c2s), K(fv_k)));
        const Val fpe = g.call2("rbl_max", g.sub(g.mul(g.call1("rbl_exp2", g.mul(es, K(pe_k2s))), K(inv_pe_den)), K(inv_pe_den)), K(0.0));
        const Val afn = g.mul(g.mul(act, fl), num);
        const Val rden = g.call1("rbl_rcp", den);
        const Val F = g.mul(g.fma(afn, rden, fpe), K(d->f_max[k]));
        for (const Unit &un : units) {
            auto &s = pair_sum[un.pr];
            for (int a = 0; a < 3; ++a) { s[a] = g.fma(un.m[a], F, s[a]); s[3 + a] = g.fma(un.u[a], F, s[3 + a]); }
        }
    };
    for (int k = 0; k < nt; ++k) if (t_last[k] < 0) emit_tendon(k);      // (tendons that touch no moving link)
    for (int i = 0; i < nq; ++i) {
        g.comment("link " + std::to_string(i) + ": frame, axis, velocity");
        const int par = parent[i];
        const M3 &Rp = par < 0 ? ident : R[par];
        const V3 pp = par < 0 ? Gen::zero3() : p[par], wp = par < 0 ? Gen::zero3() : w[par], vop = par < 0 ? Gen::zero3() : vo[par];
        const Val qi = Gen::named("q[" + std::to_string(i) + "]"), qdi = Gen::named("qd[" + std::to_string(i) + "]");
        const Val sn = g.call1("rbl_sin", qi), cs = g.call1("rbl_cos", qi);
        const double *ax = d->axis + 3 * i;
        const Val oc = g.sub(K(1.0), cs);
        // exp(q [a]x) = cos I + sin [a]x + (1 - cos) a a^T; the diagonal as a^2 + cos (1 - a^2): exact for unit axes
        M3 rot;
        for (int r = 0; r < 3; ++r)
            for (int c = 0; c < 3; ++c) {
                if (r == c) { rot[3 * r + c] = g.fma(cs, K(1.0 - ax[r] * ax[r]), K(ax[r] * ax[r])); continue; }
                const int o3 = 3 - r - c;                                       // the third index
                const double sgn = ((r + 1) % 3 == c) ? -1.0 : 1.0;             // [a]x: (0,1) = -a2, (1,2) = -a0, (2,0) = -a1
                const Val sk = g.mul(sn, K(sgn * ax[o3]));
                const Val ok = g.mul(oc, K(ax[r] * ax[c]));
                rot[3 * r + c] = g.add(sk, ok);
            }
        R[i] = g.matmul(Rp, rot);
        p[i] = g.vadd(pp, g.matvec(Rp, Gen::KV(d->origin + 3 * i)));
        z[i] = g.matvec(Rp, Gen::KV(ax));
        sl[i] = g.cross(p[i], z[i]);
        w[i] = g.vfma(z[i], qdi, wp);
        vo[i] = g.vfma(sl[i], qdi, vop);
        const V3 ca = g.vscale(g.cross(wp, z[i]), qdi);                          // (w_i x z_i = w_p x z_i)
        const V3 wxsl = g.cross(w[i], sl[i]);
        const V3 voxz = g.cross(vo[i], z[i]);
        const V3 cl = g.vscale(g.vadd(wxsl, voxz), qdi);
        for (int a = 0; a < 3; ++a) { cacc[i][a] = ca[a]; cacc[i][3 + a] = cl[a]; }
        for (int a = 0; a < 6; ++a) cslot[i][a] = (lds_c && !cacc[i][a].k) ? g.lds_store(cacc[i][a]) : -1;
        for (int r = 0; r < 6; ++r) for (int c = r; c < 6; ++c) Iown[i].m[r][c] = K(0.0);
        bown[i] = {K(0), K(0), K(0), K(0), K(0), K(0)};
        const double mass = d->mass[i];
        const double *I6 = d->inertia + 6 * i;
        bool massless = mass == 0.0;
        for (int a = 0; a < 6; ++a) massless = massless && I6[a] == 0.0;
        if (!massless) {
            Sym6 &I = Iown[i];
            const V3 cw = g.vadd(p[i], g.matvec(R[i], Gen::KV(d->com + 3 * i)));
            const V3 h = g.vscale(cw, K(mass));
            // I_w = R I_c R^T (I_c symmetric: xx,yy,zz,xy,xz,yz)
            const M3 Ic = {K(I6[0]), K(I6[3]), K(I6[4]), K(I6[3]), K(I6[1]), K(I6[5]), K(I6[4]), K(I6[5]), K(I6[2])};
            const M3 T = g.matmul(R[i], Ic);
            Val Io[3][3];
            for (int r = 0; r < 3; ++r)
                for (int c = r; c < 3; ++c) {
                    Val e = g.dot({{T[3 * r], R[i][3 * c]}, {T[3 * r + 1], R[i][3 * c + 1]}, {T[3 * r + 2], R[i][3 * c + 2]}});
                    if (r == c) {
                        const int n1 = (r + 1) % 3, n2 = (r + 2) % 3;
                        e = g.add(e, g.fma(h[n1], cw[n1], g.mul(h[n2], cw[n2])));
                    } else {
                        e = g.sub(e, g.mul(h[r], cw[c]));
                    }
                    Io[r][c] = e; Io[c][r] = e;
                }
            // spatial inertia about the world origin: [[Io, [h]x], [[h]x^T, m 1]]
            for (int r = 0; r < 3; ++r) for (int c = r; c < 3; ++c) I.m[r][c] = Io[r][c];
            const Val hx[3][3] = {{K(0), Gen::negv(h[2]), h[1]}, {h[2], K(0), Gen::negv(h[0])}, {Gen::negv(h[1]), h[0], K(0)}};
            for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) I.m[r][3 + c] = hx[r][c];
            for (int r = 0; r < 3; ++r) I.m[3 + r][3 + r] = K(mass);
            // bias force v x* (I v)
            const V3 Iow = {g.dot({{Io[0][0], w[i][0]}, {Io[0][1], w[i][1]}, {Io[0][2], w[i][2]}}),
                            g.dot({{Io[1][0], w[i][0]}, {Io[1][1], w[i][1]}, {Io[1][2], w[i][2]}}),
                            g.dot({{Io[2][0], w[i][0]}, {Io[2][1], w[i][1]}, {Io[2][2], w[i][2]}})};
            const V3 hxvo = g.cross(h, vo[i]);
            const V3 Iva = g.vadd(Iow, hxvo);
            const V3 mvo = g.vscale(vo[i], K(mass));
            const V3 hxw = g.cross(h, w[i]);
            const V3 Ivl = g.vsub(mvo, hxw);
            const V3 wxIa = g.cross(w[i], Iva);
            const V3 voxIl = g.cross(vo[i], Ivl);
            const V3 ba = g.vadd(wxIa, voxIl);
            const V3 bl = g.cross(w[i], Ivl);
            for (int a = 0; a < 3; ++a) { bown[i][a] = ba[a]; bown[i][3 + a] = bl[a]; }
        }
        for (int k = 0; k < nt; ++k) if (t_last[k] == i) emit_tendon(k);
        g.barrier();
    }
    auto load_c = [&](int i) {
        std::array<Val, 6> c = cacc[i];
        for (int a = 0; a < 6; ++a) if (cslot[i][a] >= 0) c[a] = g.lds_load(cslot[i][a]);
        return c;
    };

    // pT = -f_ext: the crossing la -> lb pulls la towards lb (f_ext_la += W, f_ext_lb -= W)
    std::vector<std::array<Val, 6>> pT(nq);
    for (int i = 0; i < nq; ++i) pT[i] = {K(0), K(0), K(0), K(0), K(0), K(0)};
    for (const auto &pr : pair_order) {
        const auto &s = pair_sum[pr];
        for (int a = 0; a < 6; ++a) {
            if (pr.first >= 0) pT[pr.first][a] = g.sub(pT[pr.first][a], s[a]);
            if (pr.second >= 0) pT[pr.second][a] = g.add(pT[pr.second][a], s[a]);
        }
    }
    // ---------------- sweep 2: articulated inertias and bias forces, leaves to root ----------------
    std::vector<Sym6> IA(nq);
    std::vector<std::array<Val, 6>> pA(nq), U(nq);
    std::vector<Val> invD(nq), uu(nq);
    for (int i = 0; i < nq; ++i) {
        for (int r = 0; r < 6; ++r) for (int c = r; c < 6; ++c) IA[i].m[r][c] = K(0.0);
        for (int a = 0; a < 6; ++a) pA[i][a] = g.add(bown[i][a], pT[i][a]);
    }
    // the parked c of a link is requested one link ahead of its use: a wave is alone on its SIMD, so an LDS latency
    // met at the point of use is idle time (the scheduling barriers keep the request where it is written)
    std::vector<std::array<Val, 6>> cpre(nq);
    std::vector<char> cpre_ok(nq, 0);
    auto prefetch_c = [&](int i) { if (i >= 0 && i < nq && !cpre_ok[i]) { cpre[i] = load_c(i); cpre_ok[i] = 1; } };
    auto next_user = [&](int i) { int n = i - 1; while (n >= 0 && parent[n] < 0) --n; return n; };   // (a root's backward step uses no c)
    auto take_c = [&](int i) { if (!cpre_ok[i]) cpre[i] = load_c(i); cpre_ok[i] = 0; return cpre[i]; };
    prefetch_c(next_user(nq));
    for (int i = nq - 1; i >= 0; --i) {
        g.comment("link " + std::to_string(i) + ": backward pass");
        prefetch_c(next_user(i));
        Sym6 &I = IA[i];
        for (int r = 0; r < 6; ++r) for (int c = r; c < 6; ++c) I.m[r][c] = g.add(I.m[r][c], Iown[i].m[r][c]);
        const std::array<Val, 6> s = {z[i][0], z[i][1], z[i][2], sl[i][0], sl[i][1], sl[i][2]};
        for (int r = 0; r < 6; ++r) {
            std::vector<std::pair<Val, Val>> terms;
            for (int c = 0; c < 6; ++c) terms.push_back({I.at(r, c), s[c]});
            U[i][r] = g.dot(terms);
        }
        std::vector<std::pair<Val, Val>> sU, sP;
        for (int r = 0; r < 6; ++r) { sU.push_back({s[r], U[i][r]}); sP.push_back({s[r], pA[i][r]}); }
        const Val D = g.add(g.dot(sU), K(d->armature[i]));
        invD[i] = g.call1("rbl_rcp", D);
        const Val dqd = g.mul(Gen::named("qd[" + std::to_string(i) + "]"), K(d->damping[i]));
        const Val spa = g.dot(sP);
        uu[i] = Gen::negv(g.add(dqd, spa));
        const int par = parent[i];
        if (par >= 0) {
            // I^a = I^A - U U^T / D,  p^a = p^A + I^a c + U u / D, added to the parent
            const std::array<Val, 6> c = take_c(i);
            std::array<Val, 6> Kk;
            for (int r = 0; r < 6; ++r) Kk[r] = g.mul(U[i][r], invD[i]);
            Sym6 Ia;
            for (int r = 0; r < 6; ++r) for (int cc = r; cc < 6; ++cc) Ia.m[r][cc] = g.sub(I.m[r][cc], g.mul(Kk[r], U[i][cc]));
            const Val ud = g.mul(uu[i], invD[i]);
            for (int r = 0; r < 6; ++r) {
                std::vector<std::pair<Val, Val>> terms;
                for (int cc = 0; cc < 6; ++cc) terms.push_back({Ia.at(r, cc), c[cc]});
                const Val pu = g.fma(U[i][r], ud, pA[i][r]);
                const Val pa = g.add(pu, g.dot(terms));
                pA[par][r] = g.add(pA[par][r], pa);
            }
            for (int r = 0; r < 6; ++r) for (int cc = r; cc < 6; ++cc) IA[par].m[r][cc] = g.add(IA[par].m[r][cc], Ia.m[r][cc]);
        }
        g.barrier();
    }
    // ---------------- sweep 3: accelerations, root to leaves ----------------
    std::vector<std::array<Val, 6>> acc(nq);
    const std::array<Val, 6> a0 = {K(0), K(0), K(0), K(-d->gravity[0]), K(-d->gravity[1]), K(-d->gravity[2])};
    for (int i = 0; i < nq; ++i) cpre_ok[i] = 0;
    prefetch_c(0);
    for (int i = 0; i < nq; ++i) {
        g.comment("link " + std::to_string(i) + ": acceleration");
        prefetch_c(i + 1);
        const std::array<Val, 6> &apar = parent[i] < 0 ? a0 : acc[parent[i]];
        const std::array<Val, 6> c = take_c(i);
        std::array<Val, 6> ap;
        for (int r = 0; r < 6; ++r) ap[r] = g.add(apar[r], c[r]);
        std::vector<std::pair<Val, Val>> terms;
        for (int r = 0; r < 6; ++r) terms.push_back({U[i][r], ap[r]});
        const Val qdd = g.mul(g.sub(uu[i], g.dot(terms)), invD[i]);
        g.store("qdd[" + std::to_string(i) + "]", qdd);
        if (!children[i].empty()) {
            const std::array<Val, 6> s = {z[i][0], z[i][1], z[i][2], sl[i][0], sl[i][1], sl[i][2]};
            for (int r = 0; r < 6; ++r) acc[i][r] = g.fma(s[r], qdd, ap[r]);
        }
        g.barrier();
    }

    // ---------------- the header ----------------
    std::string t;
    char buf[256];
    t += "// GENERATED by gym_roboy_amd/csrc/tree_lane_gen.hpp - do not edit; the acceleration of ONE robot as straight-line code.\n";
    std::snprintf(buf, sizeof buf, "#define RBL_NQ %d\n#define RBL_NT %d\n#define RBL_ACCEL_LDS %d\n", nq, nt, g.n_lds);
    t += buf;
    auto table = [&](const char *name, int n, auto value) {
        t += std::string("RBL_TABLE(") + name + ", " + std::to_string(n) + ") = {";
        for (int k = 0; k < n; ++k) t += Gen::lit(value(k)) + (k + 1 < n ? ", " : "");
        t += "};\n";
    };
    t += "namespace RBL_NS {\n";
    table("KSG", nt, [&](int k) { return ksg[k]; });
    table("QLO", nq, [&](int k) { return d->q_lo[k]; });
    table("QHI", nq, [&](int k) { return d->q_hi[k]; });
    table("VMAX", nq, [&](int k) { return d->qd_max[k]; });
    t += "template <class RBL_L>\nRBL_FN void rbl_accel(const float (&q)[RBL_NQ], const float (&qd)[RBL_NQ], const float (&spu)[RBL_NT], "
         "float (&qdd)[RBL_NQ], RBL_L rbl_lds) {\n";
    t += g.body(out.n_stmt, out.flops, out.max_live);
    t += "}\n}  // namespace RBL_NS\n";
    out.text = t;
    out.n_q = nq; out.n_t = nt; out.lds_slots = g.n_lds;
    out.hash = fnv1a(t);
    return RB_OK;
}

// LDS slots per lane of the kernels of tree_lane.hpp (the formula of its LDS_SLOTS): the acceleration's slots, aliased
// by the row transposes, then the RK4 accumulators
inline int lane_lds_slots(const Generated &g) {
    const int stage = 5 * g.n_q > 3 * g.n_q + g.n_t ? 5 * g.n_q : 3 * g.n_q + g.n_t;
    return (g.lds_slots > stage ? g.lds_slots : stage) + 2 * g.n_q;
}
inline size_t lane_lds_bytes_per_wave(const Generated &g) { return size_t(lane_lds_slots(g)) * 64 * 4; }

}  // namespace rblg
