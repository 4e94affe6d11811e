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
