"""TEST INFRASTRUCTURE - fp64 numpy restatement of the tendon-robot physics step.

*** parity unpinned ***  The reference ships no physics: its step is an RPC
(``/root/reference/gym_roboy/envs/simulations/ros_simulation_client.py:48-60``)
into the external CARDSflow simulator, which is neither vendored nor pinned
(``/root/reference/README.md:34-36``), and its tests hold no golden vector for
the step (``gym_roboy/envs/tests/test_simulation_client.py:13-76`` is
qualitative).  This file therefore *is* the specification of the model named
in BASELINE.json's ``north_star`` (via-point routing, cable-length Jacobian,
Hill-type muscle, joint-space M^-1 tau, semi-implicit Euler / RK4); the HIP
kernels are checked against it, not against CARDSflow.  What the reference
does constrain is cited at each place it applies.

Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import this module.  The product never does.

Written for obviousness, not speed: textbook Jacobian-sum mass matrix,
recursive Newton-Euler bias, dense solve; everything batched over the leading
env axis and looped in Python over joints / tendons / via-points.
"""
import numpy as np

EULER = 0   # semi-implicit (symplectic) Euler
RK4 = 1


def _cross(a, b):
    return np.cross(a, b)


def _rodrigues(axis, angle):
    """Rotation matrices [N,3,3] for a fixed unit ``axis`` and angles [N]."""
    x, y, z = axis
    K = np.array([[0.0, -z, y], [z, 0.0, -x], [-y, x, 0.0]])
    s = np.sin(angle)[:, None, None]
    c = np.cos(angle)[:, None, None]
    return np.eye(3)[None] + s * K[None] + (1.0 - c) * (K @ K)[None]


def _sym6_to_mat(i6):
    xx, yy, zz, xy, xz, yz = i6
    return np.array([[xx, xy, xz], [xy, yy, yz], [xz, yz, zz]])


class TendonRobotOracle:
    """fp64 model of one robot description, evaluated for a batch of envs."""

    def __init__(self, desc):
        self.n_q, self.n_t = int(desc.n_q), int(desc.n_t)
        self.parent = np.asarray(desc.parent, dtype=np.int64)
        self.axis = np.asarray(desc.axis, dtype=np.float64)
        self.origin = np.asarray(desc.origin, dtype=np.float64)
        self.mass = np.asarray(desc.mass, dtype=np.float64)
        self.com = np.asarray(desc.com, dtype=np.float64)
        self.inertia = [_sym6_to_mat(i6) for i6 in np.asarray(desc.inertia, dtype=np.float64)]
        self.armature = np.asarray(desc.armature, dtype=np.float64)
        self.damping = np.asarray(desc.damping, dtype=np.float64)
        self.q_lo = np.asarray(desc.q_lo, dtype=np.float64)
        self.q_hi = np.asarray(desc.q_hi, dtype=np.float64)
        self.qd_max = np.asarray(desc.qd_max, dtype=np.float64)
        self.gravity = np.asarray(desc.gravity, dtype=np.float64)
        self.vp_offset = np.asarray(desc.vp_offset, dtype=np.int64)
        self.vp_link = np.asarray(desc.vp_link, dtype=np.int64)
        self.vp_pos = np.asarray(desc.vp_pos, dtype=np.float64)
        self.f_max = np.asarray(desc.f_max, dtype=np.float64)
        m = desc.muscle
        self.kp, self.sigma, self.v_max = m["kp"], m["setpoint_scale"], m["v_max"]
        self.fl_width, self.kpe, self.e0 = m["fl_width"], m["kpe"], m["e0"]
        # force-velocity curve: one rational (1+c1 v)/(1+c2 v) per branch,
        # C1-continuous at v=0
        a, n = m["fv_a"], m["fv_n"]
        self.fv_short = (1.0, -1.0 / a)
        slope0 = 1.0 + 1.0 / a
        c2 = slope0 / (n - 1.0)
        self.fv_long = (n * c2, c2)
        # anc[i, j]: joint j lies on the path base -> link i (j == i included)
        self.anc = np.zeros((self.n_q, self.n_q), dtype=bool)
        for i in range(self.n_q):
            j = i
            while j >= 0:
                self.anc[i, j] = True
                j = self.parent[j]
        # rest length = tendon length in the zero pose; also the optimal
        # fibre length of the muscle (normalised length 1 at q = 0)
        self.l0 = self.tendon_geometry(np.zeros((1, self.n_q)))[0][0].copy()

    # ------------------------------------------------------------ kinematics
    def kinematics(self, q):
        """World rotation R[N,nq,3,3], joint origin p[N,nq,3], joint axis z[N,nq,3]."""
        N = q.shape[0]
        R = np.zeros((N, self.n_q, 3, 3))
        p = np.zeros((N, self.n_q, 3))
        z = np.zeros((N, self.n_q, 3))
        eye = np.broadcast_to(np.eye(3), (N, 3, 3))
        for i in range(self.n_q):
            par = self.parent[i]
            Rp = eye if par < 0 else R[:, par]
            pp = 0.0 if par < 0 else p[:, par]
            p[:, i] = pp + Rp @ self.origin[i]
            z[:, i] = Rp @ self.axis[i]
            R[:, i] = Rp @ _rodrigues(self.axis[i], q[:, i])
        return R, p, z

    def _via_point(self, v, R, p, z):
        """World position x[N,3] and Jacobian J[N,3,nq] of via-point ``v``."""
        N = R.shape[0]
        link = self.vp_link[v]
        J = np.zeros((N, 3, self.n_q))
        if link < 0:
            return np.broadcast_to(self.vp_pos[v], (N, 3)).copy(), J
        x = p[:, link] + R[:, link] @ self.vp_pos[v]
        for j in range(self.n_q):
            if self.anc[link, j]:
                J[:, :, j] = _cross(z[:, j], x - p[:, j])
        return x, J

    def tendon_geometry(self, q):
        """Tendon lengths l[N,nt] and cable-length Jacobian L = dl/dq [N,nt,nq]."""
        R, p, z = self.kinematics(q)
        N = q.shape[0]
        length = np.zeros((N, self.n_t))
        L = np.zeros((N, self.n_t, self.n_q))
        for k in range(self.n_t):
            v0, v1 = self.vp_offset[k], self.vp_offset[k + 1]
            xa, Ja = self._via_point(v0, R, p, z)
            for v in range(v0 + 1, v1):
                xb, Jb = self._via_point(v, R, p, z)
                d = xb - xa
                seg = np.sqrt(np.sum(d * d, axis=1))
                u = d / seg[:, None]
                length[:, k] += seg
                L[:, k, :] += np.einsum("nc,ncj->nj", u, Jb - Ja)
                xa, Ja = xb, Jb
        return length, L

    # ---------------------------------------------------------------- muscle
    def muscle_force(self, length, length_rate, setpoint):
        """Hill-type tendon force F[N,nt] >= 0 (tendons only pull)."""
        l0 = self.l0
        err = (length - l0 - self.sigma * setpoint) / l0
        act = np.clip(self.kp * err, 0.0, 1.0)          # saturating P-law
        ln = length / l0
        f_l = np.exp(-((ln - 1.0) / self.fl_width) ** 2)
        v = length_rate / (self.v_max * l0)              # >0: lengthening
        c1 = np.where(v > 0.0, self.fv_long[0], self.fv_short[0])
        c2 = np.where(v > 0.0, self.fv_long[1], self.fv_short[1])
        v = np.maximum(v, -1.0)
        f_v = np.maximum((1.0 + c1 * v) / (1.0 + c2 * v), 0.0)
        f_pe = np.maximum(
            (np.exp(self.kpe * (ln - 1.0) / self.e0) - 1.0) / (np.exp(self.kpe) - 1.0), 0.0)
        return self.f_max * (act * f_l * f_v + f_pe)

    # -------------------------------------------------------------- dynamics
    def mass_matrix(self, q):
        """M(q) = sum_i m_i Jv_i^T Jv_i + Jw_i^T (R_i I_i R_i^T) Jw_i + diag(armature)."""
        R, p, z = self.kinematics(q)
        N = q.shape[0]
        M = np.zeros((N, self.n_q, self.n_q))
        for i in range(self.n_q):
            if self.mass[i] == 0.0 and not np.any(self.inertia[i]):
                continue
            c = p[:, i] + R[:, i] @ self.com[i]
            Jv = np.zeros((N, 3, self.n_q))
            Jw = np.zeros((N, 3, self.n_q))
            for j in range(self.n_q):
                if self.anc[i, j]:
                    Jv[:, :, j] = _cross(z[:, j], c - p[:, j])
                    Jw[:, :, j] = z[:, j]
            Iw = R[:, i] @ self.inertia[i] @ np.swapaxes(R[:, i], 1, 2)
            M += self.mass[i] * np.swapaxes(Jv, 1, 2) @ Jv
            M += np.swapaxes(Jw, 1, 2) @ Iw @ Jw
        M += np.diag(self.armature)[None]
        return M

    def bias(self, q, qd):
        """Coriolis + centrifugal + gravity torques: recursive Newton-Euler with
        zero joint acceleration and base acceleration -g."""
        R, p, z = self.kinematics(q)
        N = q.shape[0]
        nq = self.n_q
        w = np.zeros((N, nq, 3)); al = np.zeros((N, nq, 3)); ap = np.zeros((N, nq, 3))
        f = np.zeros((N, nq, 3)); n = np.zeros((N, nq, 3))
        zero = np.zeros((N, 3))
        for i in range(nq):
            par = self.parent[i]
            if par < 0:
                w_p, al_p, a_pp, p_p = zero, zero, -self.gravity[None] + zero, zero
            else:
                w_p, al_p, a_pp, p_p = w[:, par], al[:, par], ap[:, par], p[:, par]
            r = p[:, i] - p_p
            ap[:, i] = a_pp + _cross(al_p, r) + _cross(w_p, _cross(w_p, r))
            w[:, i] = w_p + z[:, i] * qd[:, i, None]
            al[:, i] = al_p + _cross(w_p, z[:, i]) * qd[:, i, None]
            rc = R[:, i] @ self.com[i]
            ac = ap[:, i] + _cross(al[:, i], rc) + _cross(w[:, i], _cross(w[:, i], rc))
            F = self.mass[i] * ac
            Iw = R[:, i] @ self.inertia[i] @ np.swapaxes(R[:, i], 1, 2)
            Iw_w = np.einsum("nab,nb->na", Iw, w[:, i])
            Nn = np.einsum("nab,nb->na", Iw, al[:, i]) + _cross(w[:, i], Iw_w)
            f[:, i] = F
            n[:, i] = Nn + _cross(rc, F)
        tau = np.zeros((N, nq))
        for i in range(nq - 1, -1, -1):
            tau[:, i] = np.sum(z[:, i] * n[:, i], axis=1)
            par = self.parent[i]
            if par >= 0:
                f[:, par] += f[:, i]
                n[:, par] += n[:, i] + _cross(p[:, i] - p[:, par], f[:, i])
        return tau

    def acceleration(self, q, qd, setpoint):
        """qdd = M^-1 ( -L^T F - D qd - bias )."""
        length, L = self.tendon_geometry(q)
        length_rate = np.einsum("nkj,nj->nk", L, qd)
        F = self.muscle_force(length, length_rate, setpoint)
        tau = -np.einsum("nkj,nk->nj", L, F) - self.damping * qd - self.bias(q, qd)
        return np.linalg.solve(self.mass_matrix(q), tau[:, :, None])[:, :, 0]

    # ------------------------------------------------------------ integrator
    def _limit(self, q, qd):
        """Velocity saturation, then joint limits: clamp the angle, drop the
        velocity component that pushes outward, flag the env infeasible.  No
        reset; pushing on keeps it infeasible
        (``test_simulation_client.py:54-68``)."""
        qd = np.clip(qd, -self.qd_max, self.qd_max)
        over, under = q > self.q_hi, q < self.q_lo
        q = np.where(over, self.q_hi, np.where(under, self.q_lo, q))
        qd = np.where(over, np.minimum(qd, 0.0), np.where(under, np.maximum(qd, 0.0), qd))
        return q, qd, ~np.any(over | under, axis=1)

    def step(self, q, qd, setpoint, step_size=0.1, integrator=EULER, n_substeps=1):
        """One env step of ``step_size`` seconds (``ros_simulation_client.py:22``)
        with the set-points held.  Returns (q', qd', feasible[N] bool)."""
        q = np.array(q, dtype=np.float64).reshape(-1, self.n_q)
        qd = np.array(qd, dtype=np.float64).reshape(-1, self.n_q)
        s = np.asarray(setpoint, dtype=np.float64).reshape(-1, self.n_t)
        feasible = np.ones(q.shape[0], dtype=bool)
        h = step_size / n_substeps
        for _ in range(n_substeps):
            if integrator == EULER:
                qd = qd + h * self.acceleration(q, qd, s)
                qd = np.clip(qd, -self.qd_max, self.qd_max)
                q = q + h * qd
            elif integrator == RK4:
                # stage velocities saturate at the actuator speed limit, so a
                # step never moves a joint further than h * qd_max
                sat = lambda v: np.clip(v, -self.qd_max, self.qd_max)
                k1q = sat(qd)
                k1v = self.acceleration(q, k1q, s)
                k2q = sat(qd + 0.5 * h * k1v)
                k2v = self.acceleration(q + 0.5 * h * k1q, k2q, s)
                k3q = sat(qd + 0.5 * h * k2v)
                k3v = self.acceleration(q + 0.5 * h * k2q, k3q, s)
                k4q = sat(qd + h * k3v)
                k4v = self.acceleration(q + h * k3q, k4q, s)
                q = q + (h / 6.0) * (k1q + 2.0 * k2q + 2.0 * k3q + k4q)
                qd = qd + (h / 6.0) * (k1v + 2.0 * k2v + 2.0 * k3v + k4v)
            else:
                raise ValueError("unknown integrator %r" % (integrator,))
            q, qd, ok = self._limit(q, qd)
            feasible &= ok
        return q, qd, feasible

    # ---------------------------------------------------------- diagnostics
    def total_energy(self, q, qd):
        """Kinetic + gravitational potential energy (conservation check)."""
        R, p, _ = self.kinematics(q)
        M = self.mass_matrix(q)
        ke = 0.5 * np.einsum("ni,nij,nj->n", qd, M, qd)
        pe = np.zeros(q.shape[0])
        for i in range(self.n_q):
            c = p[:, i] + R[:, i] @ self.com[i]
            pe -= self.mass[i] * (c @ self.gravity)
        return ke + pe
