/* TEST INFRASTRUCTURE - plain C restatement of the tendon-robot physics step.
 *
 * *** parity unpinned ***  The reference contains no physics (its step is an
 * RPC into the external CARDSflow simulator,
 * gym_roboy/envs/simulations/ros_simulation_client.py:48-60, README.md:34-36)
 * and no golden vector for it (gym_roboy/envs/tests/test_simulation_client.py:13-76
 * is qualitative), so this file restates the build's own model spec
 *  exactly as oracle/physics_np.py does: generic joint tree,
 * via-point routing, cable-length Jacobian, Hill-type muscle, Jacobian-sum
 * mass matrix, recursive Newton-Euler bias, Cholesky solve, semi-implicit
 * Euler / RK4.  It is the second witness of the spec (C vs numpy agree to
 * rounding) and the CPU baseline timed by bench.py ("port").
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library.  The product (gym_roboy_amd/) never does.
 *
 * Build: make -C oracle  ->  liboracle_f64.so, liboracle_f32.so (-DORC_F32).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "../include/roboy_sim.h"

#ifdef ORC_F32
typedef float real;
#define R_SIN sinf
#define R_COS cosf
#define R_EXP expf
#define R_SQRT sqrtf
#else
typedef double real;
#define R_SIN sin
#define R_COS cos
#define R_EXP exp
#define R_SQRT sqrt
#endif

#define MAXQ 32

typedef struct orc_model {
    int n_q, n_t, n_vp;
    int parent[MAXQ];
    unsigned char anc[MAXQ][MAXQ]; /* anc[i][j]: joint j on the path base -> link i */
    real axis[MAXQ][3], origin[MAXQ][3], mass[MAXQ], com[MAXQ][3], inertia[MAXQ][3][3];
    real armature[MAXQ], damping[MAXQ], q_lo[MAXQ], q_hi[MAXQ], qd_max[MAXQ], gravity[3];
    int *vp_offset, *vp_link;
    real (*vp_pos)[3];
    real *f_max, *l0;
    real kp, sigma, v_max, fl_width, kpe, e0, fv_c1s, fv_c2s, fv_c1l, fv_c2l, pe_den;
} orc_model;

static void cross3(const real a[3], const real b[3], real o[3]) {
    o[0] = a[1] * b[2] - a[2] * b[1];
    o[1] = a[2] * b[0] - a[0] * b[2];
    o[2] = a[0] * b[1] - a[1] * b[0];
}
static void matvec(const real m[3][3], const real v[3], real o[3]) {
    for (int r = 0; r < 3; ++r) o[r] = m[r][0] * v[0] + m[r][1] * v[1] + m[r][2] * v[2];
}
static void matmul(const real a[3][3], const real b[3][3], real o[3][3]) {
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) o[r][c] = a[r][0] * b[0][c] + a[r][1] * b[1][c] + a[r][2] * b[2][c];
}
static real dot3(const real a[3], const real b[3]) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }

/* Rodrigues: I + sin K + (1 - cos) K^2 */
static void rodrigues(const real ax[3], real ang, real o[3][3]) {
    const real s = R_SIN(ang), c = R_COS(ang), x = ax[0], y = ax[1], z = ax[2];
    const real K[3][3] = {{0, -z, y}, {z, 0, -x}, {-y, x, 0}};
    real K2[3][3];
    matmul(K, K, K2);
    for (int r = 0; r < 3; ++r)
        for (int cc = 0; cc < 3; ++cc) o[r][cc] = (r == cc ? (real)1 : (real)0) + s * K[r][cc] + ((real)1 - c) * K2[r][cc];
}

typedef struct {
    real R[MAXQ][3][3], p[MAXQ][3], z[MAXQ][3];
} kin_t;

static void kinematics(const orc_model *m, const real *q, kin_t *k) {
    static const real eye[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
    for (int i = 0; i < m->n_q; ++i) {
        const int par = m->parent[i];
        const real(*Rp)[3] = par < 0 ? eye : k->R[par];
        real t[3], rot[3][3];
        matvec(Rp, m->origin[i], t);
        for (int a = 0; a < 3; ++a) k->p[i][a] = (par < 0 ? (real)0 : k->p[par][a]) + t[a];
        matvec(Rp, m->axis[i], k->z[i]);
        rodrigues(m->axis[i], q[i], rot);
        matmul(Rp, rot, k->R[i]);
    }
}

/* world position and Jacobian columns of via-point v */
static void via_point(const orc_model *m, const kin_t *k, int v, real x[3], real J[MAXQ][3]) {
    const int link = m->vp_link[v];
    for (int j = 0; j < m->n_q; ++j) J[j][0] = J[j][1] = J[j][2] = 0;
    if (link < 0) {
        for (int a = 0; a < 3; ++a) x[a] = m->vp_pos[v][a];
        return;
    }
    real t[3];
    matvec(k->R[link], m->vp_pos[v], t);
    for (int a = 0; a < 3; ++a) x[a] = k->p[link][a] + t[a];
    for (int j = 0; j < m->n_q; ++j)
        if (m->anc[link][j]) {
            real r[3] = {x[0] - k->p[j][0], x[1] - k->p[j][1], x[2] - k->p[j][2]};
            cross3(k->z[j], r, J[j]);
        }
}

/* tendon lengths len[n_t] and cable-length Jacobian L[n_t][n_q] */
static void tendon_geometry(const orc_model *m, const kin_t *k, real *len, real *L) {
    real xa[3], xb[3], Ja[MAXQ][3], Jb[MAXQ][3];
    for (int t = 0; t < m->n_t; ++t) {
        len[t] = 0;
        for (int j = 0; j < m->n_q; ++j) L[t * m->n_q + j] = 0;
        const int v0 = m->vp_offset[t], v1 = m->vp_offset[t + 1];
        via_point(m, k, v0, xa, Ja);
        for (int v = v0 + 1; v < v1; ++v) {
            via_point(m, k, v, xb, Jb);
            real d[3] = {xb[0] - xa[0], xb[1] - xa[1], xb[2] - xa[2]};
            const real seg = R_SQRT(dot3(d, d));
            const real u[3] = {d[0] / seg, d[1] / seg, d[2] / seg};
            len[t] += seg;
            for (int j = 0; j < m->n_q; ++j) {
                real dj[3] = {Jb[j][0] - Ja[j][0], Jb[j][1] - Ja[j][1], Jb[j][2] - Ja[j][2]};
                L[t * m->n_q + j] += dot3(u, dj);
            }
            memcpy(xa, xb, sizeof xa);
            memcpy(Ja, Jb, sizeof(real) * 3 * m->n_q);
        }
    }
}

static real muscle_force(const orc_model *m, int t, real len, real rate, real sp) {
    const real l0 = m->l0[t];
    const real err = (len - l0 - m->sigma * sp) / l0;
    real act = m->kp * err;
    act = act < 0 ? 0 : (act > 1 ? 1 : act);
    const real ln = len / l0, e = (ln - 1) / m->fl_width;
    const real fl = R_EXP(-(e * e));
    real v = rate / (m->v_max * l0);
    const real c1 = v > 0 ? m->fv_c1l : m->fv_c1s, c2 = v > 0 ? m->fv_c2l : m->fv_c2s;
    if (v < -1) v = -1;
    real fv = (1 + c1 * v) / (1 + c2 * v);
    if (fv < 0) fv = 0;
    real fpe = (R_EXP(m->kpe * (ln - 1) / m->e0) - 1) / m->pe_den;
    if (fpe < 0) fpe = 0;
    return m->f_max[t] * (act * fl * fv + fpe);
}

static void mass_matrix(const orc_model *m, const kin_t *k, real M[MAXQ][MAXQ]) {
    const int nq = m->n_q;
    for (int a = 0; a < nq; ++a)
        for (int b = 0; b < nq; ++b) M[a][b] = a == b ? m->armature[a] : 0;
    for (int i = 0; i < nq; ++i) {
        int has = m->mass[i] != 0;
        for (int r = 0; r < 3 && !has; ++r)
            for (int c = 0; c < 3; ++c) has |= m->inertia[i][r][c] != 0;
        if (!has) continue;
        real rc[3], c[3], Jv[MAXQ][3], Rt[3][3], tmp[3][3], Iw[3][3];
        matvec(k->R[i], m->com[i], rc);
        for (int a = 0; a < 3; ++a) c[a] = k->p[i][a] + rc[a];
        for (int j = 0; j < nq; ++j) {
            Jv[j][0] = Jv[j][1] = Jv[j][2] = 0;
            if (m->anc[i][j]) {
                real r[3] = {c[0] - k->p[j][0], c[1] - k->p[j][1], c[2] - k->p[j][2]};
                cross3(k->z[j], r, Jv[j]);
            }
        }
        for (int r = 0; r < 3; ++r)
            for (int cc = 0; cc < 3; ++cc) Rt[r][cc] = k->R[i][cc][r];
        matmul(k->R[i], m->inertia[i], tmp);
        matmul(tmp, Rt, Iw);
        for (int a = 0; a < nq; ++a) {
            if (!m->anc[i][a]) continue;
            real Iz[3];
            matvec(Iw, k->z[a], Iz);
            for (int b = 0; b < nq; ++b) {
                if (!m->anc[i][b]) continue;
                M[a][b] += m->mass[i] * dot3(Jv[a], Jv[b]) + dot3(k->z[b], Iz);
            }
        }
    }
}

/* recursive Newton-Euler with zero joint acceleration, base acceleration -g */
static void bias_forces(const orc_model *m, const kin_t *k, const real *qd, real *tau) {
    const int nq = m->n_q;
    real w[MAXQ][3], al[MAXQ][3], ap[MAXQ][3], f[MAXQ][3], n[MAXQ][3];
    for (int i = 0; i < nq; ++i) {
        const int par = m->parent[i];
        real w_p[3] = {0, 0, 0}, al_p[3] = {0, 0, 0}, a_pp[3], p_p[3] = {0, 0, 0};
        if (par < 0) {
            for (int a = 0; a < 3; ++a) a_pp[a] = -m->gravity[a];
        } else {
            memcpy(w_p, w[par], sizeof w_p); memcpy(al_p, al[par], sizeof al_p);
            memcpy(a_pp, ap[par], sizeof a_pp); memcpy(p_p, k->p[par], sizeof p_p);
        }
        real r[3] = {k->p[i][0] - p_p[0], k->p[i][1] - p_p[1], k->p[i][2] - p_p[2]};
        real t1[3], t2[3], t3[3];
        cross3(al_p, r, t1); cross3(w_p, r, t2); cross3(w_p, t2, t3);
        for (int a = 0; a < 3; ++a) ap[i][a] = a_pp[a] + t1[a] + t3[a];
        for (int a = 0; a < 3; ++a) w[i][a] = w_p[a] + k->z[i][a] * qd[i];
        cross3(w_p, k->z[i], t1);
        for (int a = 0; a < 3; ++a) al[i][a] = al_p[a] + t1[a] * qd[i];
        real rc[3], ac[3], F[3];
        matvec(k->R[i], m->com[i], rc);
        cross3(al[i], rc, t1); cross3(w[i], rc, t2); cross3(w[i], t2, t3);
        for (int a = 0; a < 3; ++a) { ac[a] = ap[i][a] + t1[a] + t3[a]; F[a] = m->mass[i] * ac[a]; }
        real Rt[3][3], tmp[3][3], Iw[3][3], Iww[3], Ial[3], gyro[3], rcF[3];
        for (int rr = 0; rr < 3; ++rr)
            for (int cc = 0; cc < 3; ++cc) Rt[rr][cc] = k->R[i][cc][rr];
        matmul(k->R[i], m->inertia[i], tmp); matmul(tmp, Rt, Iw);
        matvec(Iw, w[i], Iww); matvec(Iw, al[i], Ial);
        cross3(w[i], Iww, gyro); cross3(rc, F, rcF);
        for (int a = 0; a < 3; ++a) { f[i][a] = F[a]; n[i][a] = Ial[a] + gyro[a] + rcF[a]; }
    }
    for (int i = nq - 1; i >= 0; --i) {
        tau[i] = dot3(k->z[i], n[i]);
        const int par = m->parent[i];
        if (par >= 0) {
            real r[3] = {k->p[i][0] - k->p[par][0], k->p[i][1] - k->p[par][1], k->p[i][2] - k->p[par][2]};
            real t[3];
            cross3(r, f[i], t);
            for (int a = 0; a < 3; ++a) { f[par][a] += f[i][a]; n[par][a] += n[i][a] + t[a]; }
        }
    }
}

/* in-place Cholesky solve of the SPD system M x = b */
static void spd_solve(int n, real M[MAXQ][MAXQ], real *b) {
    for (int j = 0; j < n; ++j) {
        real d = M[j][j];
        for (int k = 0; k < j; ++k) d -= M[j][k] * M[j][k];
        d = R_SQRT(d);
        M[j][j] = d;
        for (int i = j + 1; i < n; ++i) {
            real s = M[i][j];
            for (int k = 0; k < j; ++k) s -= M[i][k] * M[j][k];
            M[i][j] = s / d;
        }
    }
    for (int i = 0; i < n; ++i) {
        real s = b[i];
        for (int k = 0; k < i; ++k) s -= M[i][k] * b[k];
        b[i] = s / M[i][i];
    }
    for (int i = n - 1; i >= 0; --i) {
        real s = b[i];
        for (int k = i + 1; k < n; ++k) s -= M[k][i] * b[k];
        b[i] = s / M[i][i];
    }
}

typedef struct { real *len, *L; } scratch_t;

static void acceleration(const orc_model *m, const real *q, const real *qd, const real *sp, real *qdd,
                         scratch_t *s) {
    const int nq = m->n_q;
    kin_t k;
    real M[MAXQ][MAXQ], bias[MAXQ];
    kinematics(m, q, &k);
    tendon_geometry(m, &k, s->len, s->L);
    for (int j = 0; j < nq; ++j) qdd[j] = 0;
    for (int t = 0; t < m->n_t; ++t) {
        real rate = 0;
        for (int j = 0; j < nq; ++j) rate += s->L[t * nq + j] * qd[j];
        const real F = muscle_force(m, t, s->len[t], rate, sp[t]);
        for (int j = 0; j < nq; ++j) qdd[j] -= s->L[t * nq + j] * F;
    }
    bias_forces(m, &k, qd, bias);
    for (int j = 0; j < nq; ++j) qdd[j] -= m->damping[j] * qd[j] + bias[j];
    mass_matrix(m, &k, M);
    spd_solve(nq, M, qdd);
}

static real clampr(real v, real lo, real hi) { return v < lo ? lo : (v > hi ? hi : v); }

/* velocity saturation + joint limits (test_simulation_client.py:54-68: clamp,
 * flag infeasible, no reset) */
static int limit(const orc_model *m, real *q, real *qd) {
    int ok = 1;
    for (int j = 0; j < m->n_q; ++j) {
        real v = clampr(qd[j], -m->qd_max[j], m->qd_max[j]);
        if (q[j] > m->q_hi[j]) { q[j] = m->q_hi[j]; if (v > 0) v = 0; ok = 0; }
        else if (q[j] < m->q_lo[j]) { q[j] = m->q_lo[j]; if (v < 0) v = 0; ok = 0; }
        qd[j] = v;
    }
    return ok;
}

static int step_one(const orc_model *m, real *q, real *qd, const real *sp, real h, int integrator, int nsub,
                    scratch_t *s) {
    const int nq = m->n_q;
    int feasible = 1;
    real a[MAXQ], k1q[MAXQ], k1v[MAXQ], k2q[MAXQ], k2v[MAXQ], k3q[MAXQ], k3v[MAXQ], k4q[MAXQ], k4v[MAXQ], qs[MAXQ], vs[MAXQ];
    for (int sub = 0; sub < nsub; ++sub) {
        if (integrator == RB_EULER) {
            acceleration(m, q, qd, sp, a, s);
            for (int j = 0; j < nq; ++j) {
                qd[j] = clampr(qd[j] + h * a[j], -m->qd_max[j], m->qd_max[j]);
                q[j] = q[j] + h * qd[j];
            }
        } else {
            const real hh = (real)0.5 * h;
            for (int j = 0; j < nq; ++j) k1q[j] = clampr(qd[j], -m->qd_max[j], m->qd_max[j]);
            acceleration(m, q, k1q, sp, k1v, s);
            for (int j = 0; j < nq; ++j) { qs[j] = q[j] + hh * k1q[j]; k2q[j] = clampr(qd[j] + hh * k1v[j], -m->qd_max[j], m->qd_max[j]); }
            acceleration(m, qs, k2q, sp, k2v, s);
            for (int j = 0; j < nq; ++j) { qs[j] = q[j] + hh * k2q[j]; k3q[j] = clampr(qd[j] + hh * k2v[j], -m->qd_max[j], m->qd_max[j]); }
            acceleration(m, qs, k3q, sp, k3v, s);
            for (int j = 0; j < nq; ++j) { qs[j] = q[j] + h * k3q[j]; k4q[j] = clampr(qd[j] + h * k3v[j], -m->qd_max[j], m->qd_max[j]); }
            acceleration(m, qs, k4q, sp, k4v, s);
            for (int j = 0; j < nq; ++j) {
                q[j] = q[j] + (h / 6) * (k1q[j] + 2 * k2q[j] + 2 * k3q[j] + k4q[j]);
                qd[j] = qd[j] + (h / 6) * (k1v[j] + 2 * k2v[j] + 2 * k3v[j] + k4v[j]);
            }
            (void)vs;
        }
        feasible &= limit(m, q, qd);
    }
    return feasible;
}

/* ------------------------------------------------------------------ C API */
void orc_destroy(orc_model *m) {
    if (!m) return;
    free(m->vp_offset); free(m->vp_link); free(m->vp_pos); free(m->f_max); free(m->l0);
    free(m);
}

int orc_real_size(void) { return (int)sizeof(real); }

int orc_create(const rb_robot_desc *d, orc_model **out) {
    if (!d || !out || d->n_q < 1 || d->n_q > MAXQ || d->n_t < 1) return RB_EINVAL;
    orc_model *m = (orc_model *)calloc(1, sizeof(orc_model));
    if (!m) return RB_ENOMEM;
    m->n_q = d->n_q; m->n_t = d->n_t; m->n_vp = d->n_vp;
    m->vp_offset = (int *)malloc(sizeof(int) * (d->n_t + 1));
    m->vp_link = (int *)malloc(sizeof(int) * d->n_vp);
    m->vp_pos = (real(*)[3])malloc(sizeof(real) * 3 * d->n_vp);
    m->f_max = (real *)malloc(sizeof(real) * d->n_t);
    m->l0 = (real *)malloc(sizeof(real) * d->n_t);
    if (!m->vp_offset || !m->vp_link || !m->vp_pos || !m->f_max || !m->l0) { orc_destroy(m); return RB_ENOMEM; }
    for (int i = 0; i < d->n_q; ++i) {
        m->parent[i] = d->parent[i];
        for (int a = 0; a < 3; ++a) {
            m->axis[i][a] = (real)d->axis[3 * i + a]; m->origin[i][a] = (real)d->origin[3 * i + a];
            m->com[i][a] = (real)d->com[3 * i + a];
        }
        const double *ic = d->inertia + 6 * i;
        const double I[3][3] = {{ic[0], ic[3], ic[4]}, {ic[3], ic[1], ic[5]}, {ic[4], ic[5], ic[2]}};
        for (int r = 0; r < 3; ++r)
            for (int c = 0; c < 3; ++c) m->inertia[i][r][c] = (real)I[r][c];
        m->mass[i] = (real)d->mass[i]; m->armature[i] = (real)d->armature[i]; m->damping[i] = (real)d->damping[i];
        m->q_lo[i] = (real)d->q_lo[i]; m->q_hi[i] = (real)d->q_hi[i]; m->qd_max[i] = (real)d->qd_max[i];
        for (int j = i; j >= 0; j = d->parent[j]) m->anc[i][j] = 1;
    }
    for (int a = 0; a < 3; ++a) m->gravity[a] = (real)d->gravity[a];
    for (int t = 0; t <= d->n_t; ++t) m->vp_offset[t] = d->vp_offset[t];
    for (int v = 0; v < d->n_vp; ++v) {
        m->vp_link[v] = d->vp_link[v];
        for (int a = 0; a < 3; ++a) m->vp_pos[v][a] = (real)d->vp_pos[3 * v + a];
    }
    for (int t = 0; t < d->n_t; ++t) m->f_max[t] = (real)d->f_max[t];
    m->kp = (real)d->kp; m->sigma = (real)d->setpoint_scale; m->v_max = (real)d->v_max;
    m->fl_width = (real)d->fl_width; m->kpe = (real)d->kpe; m->e0 = (real)d->e0;
    const double slope0 = 1.0 + 1.0 / d->fv_a, c2l = slope0 / (d->fv_n - 1.0);
    m->fv_c1s = 1; m->fv_c2s = (real)(-1.0 / d->fv_a); m->fv_c1l = (real)(d->fv_n * c2l); m->fv_c2l = (real)c2l;
    m->pe_den = (real)(exp(d->kpe) - 1.0);
    /* rest lengths: tendon lengths in the zero pose (always evaluated in fp64
     * semantics of this build's `real`) */
    {
        real q0[MAXQ] = {0};
        kin_t k;
        real *L = (real *)malloc(sizeof(real) * d->n_t * d->n_q);
        if (!L) { orc_destroy(m); return RB_ENOMEM; }
        kinematics(m, q0, &k);
        tendon_geometry(m, &k, m->l0, L);
        free(L);
    }
    *out = m;
    return RB_OK;
}

int orc_geometry(const orc_model *m, long n, const real *q, real *len, real *L) {
    kin_t k;
    for (long i = 0; i < n; ++i) {
        kinematics(m, q + i * m->n_q, &k);
        tendon_geometry(m, &k, len + i * m->n_t, L + i * m->n_t * m->n_q);
    }
    return RB_OK;
}

int orc_rest_lengths(const orc_model *m, real *l0) {
    memcpy(l0, m->l0, sizeof(real) * m->n_t);
    return RB_OK;
}

/* q, qd: [n][n_q] in/out; sp: [n][n_t]; feasible: [n].  nthreads > 1 uses
 * OpenMP over envs (static chunks). */
int orc_step(const orc_model *m, long n, real *q, real *qd, const real *sp, unsigned char *feasible,
             double step_size, int integrator, int nsub, int nthreads) {
    if (!m || n < 0 || nsub < 1) return RB_EINVAL;
    const real h = (real)(step_size / nsub);
    int failed = 0;
#pragma omp parallel num_threads(nthreads > 0 ? nthreads : 1)
    {
        scratch_t s;
        s.len = (real *)malloc(sizeof(real) * m->n_t);
        s.L = (real *)malloc(sizeof(real) * m->n_t * m->n_q);
        if (!s.len || !s.L) {
#pragma omp atomic write
            failed = 1;
        } else {
#pragma omp for schedule(static)
            for (long i = 0; i < n; ++i)
                feasible[i] = (unsigned char)step_one(m, q + i * m->n_q, qd + i * m->n_q, sp + i * m->n_t, h,
                                                      integrator, nsub, &s);
        }
        free(s.len); free(s.L);
    }
    return failed ? RB_ENOMEM : RB_OK;
}
