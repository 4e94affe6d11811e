// TEST HARNESS (not product code): compiles the product's kernel arithmetic
// (gym_roboy_amd/csrc/msj_math.hpp + msj_build.hpp) for the host with g++ so
// the closed-form derivation can be checked against the oracle without a GPU.
// The GPU parity tests remain the gate; this only shortens the debug loop.
#include <string>
#include "../../gym_roboy_amd/csrc/msj_build.hpp"

// host set-point source for the run-time-count form (the kernels use an LDS column)
template <typename T>
struct SpRow {
    const T *v;
    T operator()(int k) const { return v[k]; }
};

template <typename T>
static int run(const rb_robot_desc *d, double step_size, int nsub, int integ, long n,
               T *q, T *qd, const T *sp, unsigned char *feas) {
    std::string err;
    if (d->n_t == 8) {
        rb::MsjConst<T, 8> c;
        int rc = rb::msj_build<T, 8>(d, step_size, nsub, &c, err);
        if (rc) return rc;
        for (long i = 0; i < n; ++i) {
            bool ok = integ == 0 ? rb::MsjModel<T, 8>::template step<0>(c, q + 3 * i, qd + 3 * i, sp + 8 * i)
                                 : rb::MsjModel<T, 8>::template step<1>(c, q + 3 * i, qd + 3 * i, sp + 8 * i);
            feas[i] = ok ? 1 : 0;
        }
        return 0;
    }
    // other tendon counts: the 16-record constants with the count read at run time (UNROLL = 0)
    rb::MsjConst<T, 16> c;
    int rc = rb::msj_build<T, 16>(d, step_size, nsub, &c, err, /*exact=*/false);
    if (rc) return rc;
    const int nt = d->n_t;
    for (long i = 0; i < n; ++i) {
        T u[16];
        for (int k = 0; k < nt; ++k) u[k] = rb::MsjModel<T, 16>::prescale(c, k, sp[nt * i + k]);   // activation offsets
        const SpRow<T> row{u};
        bool ok = integ == 0 ? rb::MsjModel<T, 16>::template step_sp<0, 0>(c, q + 3 * i, qd + 3 * i, row)
                             : rb::MsjModel<T, 16>::template step_sp<1, 0>(c, q + 3 * i, qd + 3 * i, row);
        feas[i] = ok ? 1 : 0;
    }
    return 0;
}
extern "C" int hm_step_f64(const rb_robot_desc *d, double step_size, int nsub, int integ, long n,
                           double *q, double *qd, const double *sp, unsigned char *feas) {
    return run<double>(d, step_size, nsub, integ, n, q, qd, sp, feas);
}
extern "C" int hm_step_f32(const rb_robot_desc *d, double step_size, int nsub, int integ, long n,
                           float *q, float *qd, const float *sp, unsigned char *feas) {
    return run<float>(d, step_size, nsub, integ, n, q, qd, sp, feas);
}
