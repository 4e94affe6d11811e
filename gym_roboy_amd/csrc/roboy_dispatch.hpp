// roboy_dispatch.hpp - which kernel instance a call launches: ONE table.
//
// Part of roboy_sim.hip (included there, inside its anonymous namespace, after the kernels, `struct rb_sim` and the availability
// predicates; not a stand-alone header).  Every launch of the three entry kinds - the physics step (rb_step*, rb_rollout_dev),
// the fused env step (rb_env_step*) and the open-loop fused rollout (rb_rollout_fused_dev) - goes through dispatch():
//
//     resolve(handle, entry)  ->  Key (robot class, entry, kernel form, integrator, workgroup size, constants source, variant)
//     find_row(Key)           ->  Row of TABLE: the key + a launcher (one template instantiation per row)
//     row->launch(handle, Launch)
//
// RB_KERNEL_AUTO's thresholds are DATA (AUTO_RULES): the first rule that matches the handle's class / entry / integrator / abilities
// and batch size names the form.  Table and rules are exported through the C ABI (rb_dispatch_rows, rb_auto_rules,
// rb_dispatch_current), so tests enumerate them instead of restating them (tests/test_dispatch_table.py: every row is reached by
// name and stepped against the oracle).  Rounds 1-5 selected among the same instances with four blocks of nested launch macros.

// ---------------------------------------------------------------------------------------------------- keys
enum : int { CLS_BALL8 = RB_CLASS_BALL8, CLS_BALLX = RB_CLASS_BALLX, CLS_TREE = RB_CLASS_TREE };
enum : int { ENTRY_STEP = RB_ENTRY_STEP, ENTRY_ENV = RB_ENTRY_ENV_STEP, ENTRY_FUSED = RB_ENTRY_FUSED_ROLLOUT };
enum : int { SRC_KERNARG = RB_SPEC_NONE, SRC_TABLE = RB_SPEC_TABLE, SRC_JIT = RB_SPEC_JIT };

struct Key {
    int cls, entry, form, integ, block, src, variant;
    // block: threads per workgroup (0: decided by the robot at run time - octet waves, split parts)
    // variant: mirror plane of the two-lanes-per-env form (0: x-z, 1: y-z); single-pass flag of the octet kernels; 0 otherwise
    bool operator==(const Key &o) const {
        return cls == o.cls && entry == o.entry && form == o.form && integ == o.integ && block == o.block && src == o.src && variant == o.variant;
    }
};

// what a launcher gets: envs [i0, i0 + cnt) of the batch on `stream`; the array arguments are those of the WHOLE batch
struct Launch {
    long i0 = 0, cnt = 0;
    hipStream_t stream = nullptr;
    const float *act = nullptr;
    float act_scale = 1.0f;                                      // step, fused rollout
    float *obs = nullptr, *reward = nullptr; uint32_t *done = nullptr;   // env step
    int ring = 0, n_steps = 0;                                   // fused rollout (act = the ring)
};
using Launcher = int (*)(rb_sim *, const Launch &);

struct Row {
    Key key;
    bool ranges;          // takes a sub-range of the batch (shifted pointers, own env count); false: whole batches only
    Launcher launch;
};

// ---------------------------------------------------------------------------------------------------- launchers: ball joints
inline Scale8 scale8(const rb_sim *s, float act_scale) {
    Scale8 us;
    for (int k = 0; k < NT8; ++k) us.v[k] = act_scale * s->c8.ten[k].ksg;
    return us;
}
inline PairMap pair_map(const rb_sim *s) {
    PairMap pm;
    for (int k = 0; k < 4; ++k) { pm.a[k] = 4 * s->pair_half[k]; pm.d[k] = 4 * (s->pair_image[k] - s->pair_half[k]); }
    return pm;
}

// one env per lane.  Small batches (<= RB_SMALL_BATCH envs) are latency-bound: one wave per workgroup spread over the CUs, tendon
// loop written out; large batches are issue-bound: 256-thread workgroups, rolled tendon loop on kernarg constants (one 16-dword
// scalar load per trip) or - constants as literals - the instances of RB_BAKED_UNROLL_* (RK4: the rolled-stages kernel, U = RS).
// A sub-range is addressed by shifted pointers: the planes keep their stride n.
template <int INTEG, int B, int U, bool BK>
int l_msj_step(rb_sim *s, const Launch &L) {
    const Scale8 us = scale8(s, L.act_scale);
    float *q = s->d_q + L.i0, *qd = s->d_qd + L.i0;
    uint32_t *feas = s->d_feas + L.i0;
    const float *act = L.act + size_t(L.i0) * NT8;
    if constexpr (U == RS)
        hipLaunchKernelGGL((msj_step_env_per_lane_rs<INTEG, B, BK>), dim3(blocks_for(L.cnt, B)), dim3(B), 0, L.stream, s->c8, q, qd, feas, act, us, s->n, L.cnt);
    else
        hipLaunchKernelGGL((msj_step_env_per_lane<INTEG, B, U, BK>), dim3(blocks_for(L.cnt, B)), dim3(B), 0, L.stream, s->c8, q, qd, feas, act, us, s->n, L.cnt);
    return RB_OK;
}
// the same instance family compiled by hiprtc on this robot's own constants (msj_jit.hpp); same parameter list
int l_msj_step_jit(rb_sim *s, const Launch &L) {
    Const8 c8 = s->c8;
    Scale8 us = scale8(s, L.act_scale);
    float *q = s->d_q + L.i0, *qd = s->d_qd + L.i0;
    uint32_t *feas = s->d_feas + L.i0;
    const float *act = L.act + size_t(L.i0) * NT8;
    long nn = s->n, cc = L.cnt;
    void *args[] = {&c8, &q, &qd, &feas, &act, &us, &nn, &cc};
    RB_HIP(hipModuleLaunchKernel(s->jit.step[s->integrator == RB_EULER ? 0 : 1], blocks_for(L.cnt, 256), 1, 1, 256, 1, 1, 0, L.stream, args, nullptr));
    return RB_OK;
}
// another tendon count (1..NTX): the closed form with a run-time trip count
template <int INTEG, int B>
int l_msj_step_nt(rb_sim *s, const Launch &L) {
    hipLaunchKernelGGL((msj_step_env_per_lane_nt<INTEG, B>), dim3(blocks_for(s->n, B)), dim3(B), 0, L.stream, s->cx, s->d_q, s->d_qd, s->d_feas, L.act, L.act_scale, s->n);
    return RB_OK;
}
// two lanes per env: 128 envs per 256-thread workgroup (64-thread workgroups up to RB_PAIR_SMALL_BATCH envs: spread over the CUs)
template <int INTEG, int B, int M, bool BK>
int l_msj_step_pair(rb_sim *s, const Launch &L) {
    const PairMap pm = pair_map(s);
    Scale4 us4;
    for (int k = 0; k < 4; ++k) us4.v[k] = L.act_scale * s->c8.ten[s->pair_half[k]].ksg;
    hipLaunchKernelGGL((msj_step_mirror_pairs<INTEG, B, M, BK>), dim3(blocks_for(2 * L.cnt, B)), dim3(B), 0, L.stream, s->c8p, pm,
                       s->d_q + L.i0, s->d_qd + L.i0, s->d_feas + L.i0, L.act + size_t(L.i0) * NT8, us4, s->n, L.cnt);
    return RB_OK;
}
// eight lanes per env (whole batches)
template <int INTEG>
int l_msj_step_octet(rb_sim *s, const Launch &L) {
    hipLaunchKernelGGL((msj_step_tendon_per_lane<INTEG>), dim3(blocks_for(s->n * NT8, 64)), dim3(64), 0, L.stream, s->c8, s->d_ten, s->d_q, s->d_qd,
                       s->d_feas, L.act, L.act_scale, s->n);
    return RB_OK;
}

// the open-loop fused rollout: the instance family of the step kernel of the same batch size (bit-identical to single steps only then)
template <int INTEG, int B, int U, bool BK>
int l_msj_fused(rb_sim *s, const Launch &L) {
    const Scale8 us = scale8(s, L.act_scale);
    hipLaunchKernelGGL((msj_rollout_fused<INTEG, B, U, BK>), dim3(blocks_for(s->n, B)), dim3(B), 0, L.stream, s->c8, s->d_q, s->d_qd, s->d_feas, L.act,
                       L.ring, L.n_steps, us, s->n);
    return RB_OK;
}
int l_msj_fused_jit(rb_sim *s, const Launch &L) {
    Const8 c8 = s->c8;
    Scale8 us = scale8(s, L.act_scale);
    long nn = s->n;
    const float *ring_ptr = L.act;
    int ring = L.ring, n_steps = L.n_steps;
    void *args[] = {&c8, &s->d_q, &s->d_qd, &s->d_feas, &ring_ptr, &ring, &n_steps, &us, &nn};
    RB_HIP(hipModuleLaunchKernel(s->jit.rollout[s->integrator == RB_EULER ? 0 : 1], blocks_for(s->n, 256), 1, 1, 256, 1, 1, 0, L.stream, args, nullptr));
    return RB_OK;
}

// fused env layer, ball joints: ONE argument behind the constants (msj_kernels.hpp: MsjEnvArgs).  SoA planes keep their stride n;
// everything else is indexed by env and shifted for a sub-range
inline MsjEnvArgs msj_env_args(rb_sim *s, const Launch &L) {
    const long i0 = L.i0;
    MsjEnvArgs a;
    a.e = s->env;
    for (int j = 0; j < 3; ++j) { a.box_lo[j] = s->box.lo[j]; a.box_hi[j] = s->box.hi[j]; }
    a.q = s->d_q + i0; a.qd = s->d_qd + i0; a.feas = s->d_feas + i0; a.goal = s->d_goal + i0; a.step_num = s->d_step_num + i0;
    a.ep_ret = s->d_ep_ret + i0; a.goal_count = s->d_goal_count + i0; a.act = L.act + i0 * s->n_t;
    a.obs = L.obs + i0 * 9; a.reward = L.reward + i0; a.done = L.done + i0;
    a.ep_sum = s->d_ep_sum + i0; a.ep_cnt = s->d_ep_cnt + i0; a.infeas_n = s->d_infeas_n + i0;
    a.n = s->n; a.cnt = L.cnt; a.seed = s->seed; a.env0 = uint64_t(s->env0) + uint64_t(i0);
    return a;
}
template <int INTEG, int B, int U, bool BK>
int l_msj_env(rb_sim *s, const Launch &L) {
    hipLaunchKernelGGL((msj_env_step_kernel<INTEG, B, U, Const8, BK>), dim3(blocks_for(L.cnt, B)), dim3(B), 0, L.stream, s->c8, msj_env_args(s, L));
    return RB_OK;
}
template <int INTEG, int B>
int l_msj_env_nt(rb_sim *s, const Launch &L) {
    hipLaunchKernelGGL((msj_env_step_kernel<INTEG, B, 0, ConstX>), dim3(blocks_for(L.cnt, B)), dim3(B), 0, L.stream, s->cx, msj_env_args(s, L));
    return RB_OK;
}
int l_msj_env_jit(rb_sim *s, const Launch &L) {
    Const8 c8 = s->c8;
    MsjEnvArgs a = msj_env_args(s, L);
    void *args[] = {&c8, &a};
    RB_HIP(hipModuleLaunchKernel(s->jit.env[s->integrator == RB_EULER ? 0 : 1], blocks_for(L.cnt, 256), 1, 1, 256, 1, 1, 0, L.stream, args, nullptr));
    return RB_OK;
}
template <int INTEG, int B, int M, bool BK>
int l_msj_env_pair(rb_sim *s, const Launch &L) {
    hipLaunchKernelGGL((msj_env_step_mirror_pairs<INTEG, B, M, BK>), dim3(blocks_for(2 * L.cnt, B)), dim3(B), 0, L.stream, s->c8p, pair_map(s), msj_env_args(s, L));
    return RB_OK;
}
template <int INTEG>
int l_msj_env_octet(rb_sim *s, const Launch &L) {
    hipLaunchKernelGGL((msj_env_step_tendon_per_lane<INTEG>), dim3(blocks_for(L.cnt * NT8, 64)), dim3(64), 0, L.stream, s->c8, s->d_ten, msj_env_args(s, L));
    return RB_OK;
}

// ---------------------------------------------------------------------------------------------------- launchers: joint trees
// (one struct argument for the env kernels: they read most of it behind the step - env_common.hpp, TreeEnvArgs; env-major rows: a
// sub-range is the same kernel on shifted pointers with its own env count, the per-env statistics planes keep their stride n)
inline rbe::TreeEnvArgs tree_env_args(rb_sim *s, const Launch &L) {
    const size_t nq = size_t(s->n_q), nt = size_t(s->n_t);
    const long i0 = L.i0;
    return rbe::TreeEnvArgs{s->env, s->box, s->d_q + i0 * nq, s->d_qd + i0 * nq, s->d_feas + i0, s->d_goal + i0 * nq, s->d_step_num + i0,
                            s->d_ep_ret + i0, s->d_goal_count + i0, L.act + i0 * nt, L.obs + i0 * 3 * nq, L.reward + i0, L.done + i0,
                            s->d_ep_sum + i0, s->d_ep_cnt + i0, s->d_infeas_n + i0, s->tree_host.dev.h, s->tree_host.dev.nsub, L.cnt,
                            s->seed, uint64_t(s->env0) + uint64_t(i0), s->n};
}
// a hiprtc-built joint-tree kernel: step kernels take (q, qd, feas, act, act_scale, h, nsub, n), env kernels one TreeEnvArgs
inline int launch_tree_module(rb_sim *s, const Launch &L, hipFunction_t fn, bool env, unsigned groups, unsigned threads, size_t lds) {
    if (env) {
        rbe::TreeEnvArgs ka = tree_env_args(s, L);
        void *args[] = {&ka};
        RB_HIP(hipModuleLaunchKernel(fn, groups, 1, 1, threads, 1, 1, unsigned(lds), L.stream, args, nullptr));
    } else {
        float *q = s->d_q + size_t(L.i0) * s->n_q, *qd = s->d_qd + size_t(L.i0) * s->n_q;
        uint32_t *feas = s->d_feas + L.i0;
        const float *act = L.act + size_t(L.i0) * s->n_t;
        float scale = L.act_scale, hh = s->tree_host.dev.h;
        int ns = s->tree_host.dev.nsub;
        long nn = L.cnt;
        void *args[] = {&q, &qd, &feas, &act, &scale, &hh, &ns, &nn};
        RB_HIP(hipModuleLaunchKernel(fn, groups, 1, 1, threads, 1, 1, unsigned(lds), L.stream, args, nullptr));
    }
    return RB_OK;
}

// octets of lanes per link, robot tables staged in LDS (robots without generated code)
template <int INTEG, bool SP, bool ENV>
int l_tree_aba(rb_sim *s, const Launch &L) {
    const int wv = s->tree_waves;
    const size_t lds = rbt::tree_lds_bytes(s->tree_host, wv);
    const long per_block = long(wv) * rbt::TREE_E, n = s->n;
    const unsigned blocks = unsigned((n + per_block - 1) / per_block);
    if constexpr (ENV)
        hipLaunchKernelGGL((rbt::tree_env_step_aba<INTEG, rbt::TREE_E, SP>), dim3(blocks), dim3(64 * wv), lds, L.stream, s->tree_host.dev, s->env, s->box,
                           s->d_q, s->d_qd, s->d_feas, s->d_goal, s->d_step_num, s->d_ep_ret, s->d_goal_count, L.act, L.obs, L.reward, L.done,
                           s->d_ep_sum, s->d_ep_cnt, s->d_infeas_n, n, s->seed, uint64_t(s->env0));
    else
        hipLaunchKernelGGL((rbt::tree_step_aba<INTEG, rbt::TREE_E, SP>), dim3(blocks), dim3(64 * wv), lds, L.stream, s->tree_host.dev, s->d_q, s->d_qd,
                           s->d_feas, L.act, L.act_scale, n);
    return RB_OK;
}
// one env per lane, one wave (64 envs) per workgroup: the LDS regions admit four per CU, one per SIMD
template <int INTEG, bool ENV>
int l_tree_lane(rb_sim *s, const Launch &L) {
    const unsigned waves = blocks_for(L.cnt, 64);
    const size_t lds = rblg::lane_lds_bytes_per_wave(s->lane_gen);
    if constexpr (ENV) {
        const rbe::TreeEnvArgs ka = tree_env_args(s, L);
        hipLaunchKernelGGL(rbl_baked::tree_lane_env_step<INTEG>, dim3(waves), dim3(64), lds, L.stream, ka);
    } else {
        hipLaunchKernelGGL(rbl_baked::tree_lane_step<INTEG>, dim3(waves), dim3(64), lds, L.stream, s->d_q + size_t(L.i0) * s->n_q, s->d_qd + size_t(L.i0) * s->n_q,
                           s->d_feas + L.i0, L.act + size_t(L.i0) * s->n_t, L.act_scale, s->tree_host.dev.h, s->tree_host.dev.nsub, L.cnt);
    }
    return RB_OK;
}
template <bool ENV>
int l_tree_lane_jit(rb_sim *s, const Launch &L) {
    return launch_tree_module(s, L, (ENV ? s->lane_env_k : s->lane_step_k).fn, ENV, blocks_for(L.cnt, 64), 64, rblg::lane_lds_bytes_per_wave(s->lane_gen));
}
// the split form: one workgroup of n_parts (+ helper) waves per 64 envs (whole batches)
template <int INTEG, bool ENV>
int l_tree_split(rb_sim *s, const Launch &L) {
    const unsigned groups = blocks_for(s->n, 64), threads = 64u * unsigned(s->split_gen.n_parts + s->split_gen.n_helpers);
    const size_t lds = split_lds_bytes(s->split_gen);
    if constexpr (ENV) {
        const rbe::TreeEnvArgs ka = tree_env_args(s, L);
        hipLaunchKernelGGL(rbl_split_baked::tree_split_env_step<INTEG>, dim3(groups), dim3(threads), lds, L.stream, ka);
    } else {
        hipLaunchKernelGGL(rbl_split_baked::tree_split_step<INTEG>, dim3(groups), dim3(threads), lds, L.stream, s->d_q, s->d_qd, s->d_feas, L.act, L.act_scale,
                           s->tree_host.dev.h, s->tree_host.dev.nsub, s->n);
    }
    return RB_OK;
}
template <bool ENV>
int l_tree_split_jit(rb_sim *s, const Launch &L) {
    return launch_tree_module(s, L, (ENV ? s->split_env_k : s->split_step_k).fn, ENV, blocks_for(s->n, 64),
                              64u * unsigned(s->split_gen.n_parts + s->split_gen.n_helpers), split_lds_bytes(s->split_gen));
}
// the lean two-part split form: two part waves per 64 envs, two workgroups per CU (whole batches; instances: roboy_sim_split2.hip)
template <int INTEG, bool ENV>
int l_tree_split2(rb_sim *s, const Launch &L) {
    if constexpr (ENV) rbs2::launch_env_step(INTEG, blocks_for(s->n, 64), L.stream, tree_env_args(s, L));
    else rbs2::launch_step(INTEG, blocks_for(s->n, 64), L.stream, s->d_q, s->d_qd, s->d_feas, L.act, L.act_scale, s->tree_host.dev.h, s->tree_host.dev.nsub, s->n);
    return RB_OK;
}
template <bool ENV>
int l_tree_split2_jit(rb_sim *s, const Launch &L) {
    return launch_tree_module(s, L, (ENV ? s->split2_env_k : s->split2_step_k).fn, ENV, blocks_for(s->n, 64), 64u * unsigned(s->split2_gen.n_parts),
                              split_lean_lds_bytes(s->split2_gen));
}

// ---------------------------------------------------------------------------------------------------- the table
constexpr int F_LANE = RB_KERNEL_ENV_PER_LANE, F_OCTET = RB_KERNEL_TENDON_PER_LANE, F_WAVE = RB_KERNEL_ENV_PER_WAVE, F_SPLIT = RB_KERNEL_ENV_PER_LANE_SPLIT,
              F_PAIR = RB_KERNEL_LANE_PAIR, F_SPLIT2 = RB_KERNEL_ENV_PER_LANE_SPLIT2;
constexpr int EU = RB_EULER, RK = RB_RK4;
// unroll factors of the large-batch env-per-lane instances (roboy_sim.hip: RB_BIG_UNROLL_*, RB_BAKED_UNROLL_*)
constexpr int UBE = RB_BAKED_UNROLL_EULER, UBR = RB_BAKED_UNROLL_RK4, UKE = RB_BIG_UNROLL_EULER, UKR = RB_BIG_UNROLL_RK4;
static_assert(RB_BIG_BLOCK_EULER == 256 && RB_BIG_BLOCK_RK4 == 256, "the table's large-batch rows are written for 256-thread workgroups");

//                 class      entry        form     integ block source       variant  ranges  launcher
const Row TABLE[] = {
    // ---- 8-tendon ball joints, physics step
    {{CLS_BALL8, ENTRY_STEP, F_LANE, EU, 64, SRC_KERNARG, 0}, true, l_msj_step<0, 64, 8, false>},
    {{CLS_BALL8, ENTRY_STEP, F_LANE, RK, 64, SRC_KERNARG, 0}, true, l_msj_step<1, 64, 8, false>},
    {{CLS_BALL8, ENTRY_STEP, F_LANE, EU, 64, SRC_TABLE, 0}, true, l_msj_step<0, 64, 8, true>},
    {{CLS_BALL8, ENTRY_STEP, F_LANE, RK, 64, SRC_TABLE, 0}, true, l_msj_step<1, 64, 8, true>},
    {{CLS_BALL8, ENTRY_STEP, F_LANE, EU, 256, SRC_KERNARG, 0}, true, l_msj_step<0, 256, UKE, false>},
    {{CLS_BALL8, ENTRY_STEP, F_LANE, RK, 256, SRC_KERNARG, 0}, true, l_msj_step<1, 256, UKR, false>},
    {{CLS_BALL8, ENTRY_STEP, F_LANE, EU, 256, SRC_TABLE, 0}, true, l_msj_step<0, 256, UBE, true>},
    {{CLS_BALL8, ENTRY_STEP, F_LANE, RK, 256, SRC_TABLE, 0}, true, l_msj_step<1, 256, UBR, true>},      // the headline kernel
    {{CLS_BALL8, ENTRY_STEP, F_LANE, EU, 256, SRC_JIT, 0}, true, l_msj_step_jit},
    {{CLS_BALL8, ENTRY_STEP, F_LANE, RK, 256, SRC_JIT, 0}, true, l_msj_step_jit},
    {{CLS_BALL8, ENTRY_STEP, F_OCTET, EU, 64, SRC_KERNARG, 0}, false, l_msj_step_octet<0>},
    {{CLS_BALL8, ENTRY_STEP, F_OCTET, RK, 64, SRC_KERNARG, 0}, false, l_msj_step_octet<1>},
    {{CLS_BALL8, ENTRY_STEP, F_PAIR, EU, 64, SRC_KERNARG, 0}, true, l_msj_step_pair<0, 64, 0, false>},
    {{CLS_BALL8, ENTRY_STEP, F_PAIR, EU, 64, SRC_KERNARG, 1}, true, l_msj_step_pair<0, 64, 1, false>},
    {{CLS_BALL8, ENTRY_STEP, F_PAIR, RK, 64, SRC_KERNARG, 0}, true, l_msj_step_pair<1, 64, 0, false>},
    {{CLS_BALL8, ENTRY_STEP, F_PAIR, RK, 64, SRC_KERNARG, 1}, true, l_msj_step_pair<1, 64, 1, false>},
    {{CLS_BALL8, ENTRY_STEP, F_PAIR, EU, 256, SRC_KERNARG, 0}, true, l_msj_step_pair<0, 256, 0, false>},
    {{CLS_BALL8, ENTRY_STEP, F_PAIR, EU, 256, SRC_KERNARG, 1}, true, l_msj_step_pair<0, 256, 1, false>},
    {{CLS_BALL8, ENTRY_STEP, F_PAIR, RK, 256, SRC_KERNARG, 0}, true, l_msj_step_pair<1, 256, 0, false>},
    {{CLS_BALL8, ENTRY_STEP, F_PAIR, RK, 256, SRC_KERNARG, 1}, true, l_msj_step_pair<1, 256, 1, false>},
    // (the ahead-of-time table is MsjRobot's, whose mirror plane is x-z: no table rows for the y-z variant)
    {{CLS_BALL8, ENTRY_STEP, F_PAIR, EU, 64, SRC_TABLE, 0}, true, l_msj_step_pair<0, 64, 0, true>},
    {{CLS_BALL8, ENTRY_STEP, F_PAIR, RK, 64, SRC_TABLE, 0}, true, l_msj_step_pair<1, 64, 0, true>},
    {{CLS_BALL8, ENTRY_STEP, F_PAIR, EU, 256, SRC_TABLE, 0}, true, l_msj_step_pair<0, 256, 0, true>},
    {{CLS_BALL8, ENTRY_STEP, F_PAIR, RK, 256, SRC_TABLE, 0}, true, l_msj_step_pair<1, 256, 0, true>},
    // ---- 8-tendon ball joints, fused env step
    {{CLS_BALL8, ENTRY_ENV, F_LANE, EU, 64, SRC_KERNARG, 0}, true, l_msj_env<0, 64, 8, false>},
    {{CLS_BALL8, ENTRY_ENV, F_LANE, RK, 64, SRC_KERNARG, 0}, true, l_msj_env<1, 64, 8, false>},
    {{CLS_BALL8, ENTRY_ENV, F_LANE, EU, 64, SRC_TABLE, 0}, true, l_msj_env<0, 64, 8, true>},
    {{CLS_BALL8, ENTRY_ENV, F_LANE, RK, 64, SRC_TABLE, 0}, true, l_msj_env<1, 64, 8, true>},
    {{CLS_BALL8, ENTRY_ENV, F_LANE, EU, 256, SRC_KERNARG, 0}, true, l_msj_env<0, 256, UKE, false>},
    {{CLS_BALL8, ENTRY_ENV, F_LANE, RK, 256, SRC_KERNARG, 0}, true, l_msj_env<1, 256, UKR, false>},
    {{CLS_BALL8, ENTRY_ENV, F_LANE, EU, 256, SRC_TABLE, 0}, true, l_msj_env<0, 256, UBE, true>},
    {{CLS_BALL8, ENTRY_ENV, F_LANE, RK, 256, SRC_TABLE, 0}, true, l_msj_env<1, 256, UBR, true>},
    {{CLS_BALL8, ENTRY_ENV, F_LANE, EU, 256, SRC_JIT, 0}, true, l_msj_env_jit},
    {{CLS_BALL8, ENTRY_ENV, F_LANE, RK, 256, SRC_JIT, 0}, true, l_msj_env_jit},
    {{CLS_BALL8, ENTRY_ENV, F_OCTET, EU, 64, SRC_KERNARG, 0}, true, l_msj_env_octet<0>},
    {{CLS_BALL8, ENTRY_ENV, F_OCTET, RK, 64, SRC_KERNARG, 0}, true, l_msj_env_octet<1>},
    {{CLS_BALL8, ENTRY_ENV, F_PAIR, EU, 64, SRC_KERNARG, 0}, true, l_msj_env_pair<0, 64, 0, false>},
    {{CLS_BALL8, ENTRY_ENV, F_PAIR, EU, 64, SRC_KERNARG, 1}, true, l_msj_env_pair<0, 64, 1, false>},
    {{CLS_BALL8, ENTRY_ENV, F_PAIR, RK, 64, SRC_KERNARG, 0}, true, l_msj_env_pair<1, 64, 0, false>},
    {{CLS_BALL8, ENTRY_ENV, F_PAIR, RK, 64, SRC_KERNARG, 1}, true, l_msj_env_pair<1, 64, 1, false>},
    {{CLS_BALL8, ENTRY_ENV, F_PAIR, EU, 256, SRC_KERNARG, 0}, true, l_msj_env_pair<0, 256, 0, false>},
    {{CLS_BALL8, ENTRY_ENV, F_PAIR, EU, 256, SRC_KERNARG, 1}, true, l_msj_env_pair<0, 256, 1, false>},
    {{CLS_BALL8, ENTRY_ENV, F_PAIR, RK, 256, SRC_KERNARG, 0}, true, l_msj_env_pair<1, 256, 0, false>},
    {{CLS_BALL8, ENTRY_ENV, F_PAIR, RK, 256, SRC_KERNARG, 1}, true, l_msj_env_pair<1, 256, 1, false>},
    {{CLS_BALL8, ENTRY_ENV, F_PAIR, EU, 64, SRC_TABLE, 0}, true, l_msj_env_pair<0, 64, 0, true>},
    {{CLS_BALL8, ENTRY_ENV, F_PAIR, RK, 64, SRC_TABLE, 0}, true, l_msj_env_pair<1, 64, 0, true>},
    {{CLS_BALL8, ENTRY_ENV, F_PAIR, EU, 256, SRC_TABLE, 0}, true, l_msj_env_pair<0, 256, 0, true>},
    {{CLS_BALL8, ENTRY_ENV, F_PAIR, RK, 256, SRC_TABLE, 0}, true, l_msj_env_pair<1, 256, 0, true>},
    // ---- 8-tendon ball joints, open-loop fused rollout (whole batches)
    {{CLS_BALL8, ENTRY_FUSED, F_LANE, EU, 64, SRC_KERNARG, 0}, false, l_msj_fused<0, 64, 8, false>},
    {{CLS_BALL8, ENTRY_FUSED, F_LANE, RK, 64, SRC_KERNARG, 0}, false, l_msj_fused<1, 64, 8, false>},
    {{CLS_BALL8, ENTRY_FUSED, F_LANE, EU, 64, SRC_TABLE, 0}, false, l_msj_fused<0, 64, 8, true>},
    {{CLS_BALL8, ENTRY_FUSED, F_LANE, RK, 64, SRC_TABLE, 0}, false, l_msj_fused<1, 64, 8, true>},
    {{CLS_BALL8, ENTRY_FUSED, F_LANE, EU, 256, SRC_KERNARG, 0}, false, l_msj_fused<0, 256, UKE, false>},
    {{CLS_BALL8, ENTRY_FUSED, F_LANE, RK, 256, SRC_KERNARG, 0}, false, l_msj_fused<1, 256, UKR, false>},
    {{CLS_BALL8, ENTRY_FUSED, F_LANE, EU, 256, SRC_TABLE, 0}, false, l_msj_fused<0, 256, UBE, true>},
    {{CLS_BALL8, ENTRY_FUSED, F_LANE, RK, 256, SRC_TABLE, 0}, false, l_msj_fused<1, 256, UBR, true>},
    {{CLS_BALL8, ENTRY_FUSED, F_LANE, EU, 256, SRC_JIT, 0}, false, l_msj_fused_jit},
    {{CLS_BALL8, ENTRY_FUSED, F_LANE, RK, 256, SRC_JIT, 0}, false, l_msj_fused_jit},
    // ---- ball joints with 1..16 tendons (run-time count)
    {{CLS_BALLX, ENTRY_STEP, F_LANE, EU, 64, SRC_KERNARG, 0}, false, l_msj_step_nt<0, 64>},
    {{CLS_BALLX, ENTRY_STEP, F_LANE, RK, 64, SRC_KERNARG, 0}, false, l_msj_step_nt<1, 64>},
    {{CLS_BALLX, ENTRY_STEP, F_LANE, EU, 256, SRC_KERNARG, 0}, false, l_msj_step_nt<0, 256>},
    {{CLS_BALLX, ENTRY_STEP, F_LANE, RK, 256, SRC_KERNARG, 0}, false, l_msj_step_nt<1, 256>},
    {{CLS_BALLX, ENTRY_ENV, F_LANE, EU, 64, SRC_KERNARG, 0}, true, l_msj_env_nt<0, 64>},
    {{CLS_BALLX, ENTRY_ENV, F_LANE, RK, 64, SRC_KERNARG, 0}, true, l_msj_env_nt<1, 64>},
    {{CLS_BALLX, ENTRY_ENV, F_LANE, EU, 256, SRC_KERNARG, 0}, true, l_msj_env_nt<0, 256>},
    {{CLS_BALLX, ENTRY_ENV, F_LANE, RK, 256, SRC_KERNARG, 0}, true, l_msj_env_nt<1, 256>},
    // ---- joint trees: octets (variant = single-pass tables), one wave per 64 envs, five-wave split, lean two-part split
    {{CLS_TREE, ENTRY_STEP, F_WAVE, EU, 0, SRC_KERNARG, 0}, false, l_tree_aba<0, false, false>},
    {{CLS_TREE, ENTRY_STEP, F_WAVE, EU, 0, SRC_KERNARG, 1}, false, l_tree_aba<0, true, false>},
    {{CLS_TREE, ENTRY_STEP, F_WAVE, RK, 0, SRC_KERNARG, 0}, false, l_tree_aba<1, false, false>},
    {{CLS_TREE, ENTRY_STEP, F_WAVE, RK, 0, SRC_KERNARG, 1}, false, l_tree_aba<1, true, false>},
    {{CLS_TREE, ENTRY_ENV, F_WAVE, EU, 0, SRC_KERNARG, 0}, false, l_tree_aba<0, false, true>},
    {{CLS_TREE, ENTRY_ENV, F_WAVE, EU, 0, SRC_KERNARG, 1}, false, l_tree_aba<0, true, true>},
    {{CLS_TREE, ENTRY_ENV, F_WAVE, RK, 0, SRC_KERNARG, 0}, false, l_tree_aba<1, false, true>},
    {{CLS_TREE, ENTRY_ENV, F_WAVE, RK, 0, SRC_KERNARG, 1}, false, l_tree_aba<1, true, true>},
    {{CLS_TREE, ENTRY_STEP, F_LANE, EU, 64, SRC_TABLE, 0}, true, l_tree_lane<0, false>},
    {{CLS_TREE, ENTRY_STEP, F_LANE, RK, 64, SRC_TABLE, 0}, true, l_tree_lane<1, false>},
    {{CLS_TREE, ENTRY_ENV, F_LANE, EU, 64, SRC_TABLE, 0}, true, l_tree_lane<0, true>},
    {{CLS_TREE, ENTRY_ENV, F_LANE, RK, 64, SRC_TABLE, 0}, true, l_tree_lane<1, true>},
    {{CLS_TREE, ENTRY_STEP, F_LANE, EU, 64, SRC_JIT, 0}, true, l_tree_lane_jit<false>},
    {{CLS_TREE, ENTRY_STEP, F_LANE, RK, 64, SRC_JIT, 0}, true, l_tree_lane_jit<false>},
    {{CLS_TREE, ENTRY_ENV, F_LANE, EU, 64, SRC_JIT, 0}, true, l_tree_lane_jit<true>},
    {{CLS_TREE, ENTRY_ENV, F_LANE, RK, 64, SRC_JIT, 0}, true, l_tree_lane_jit<true>},
    {{CLS_TREE, ENTRY_STEP, F_SPLIT, EU, 0, SRC_TABLE, 0}, false, l_tree_split<0, false>},
    {{CLS_TREE, ENTRY_STEP, F_SPLIT, RK, 0, SRC_TABLE, 0}, false, l_tree_split<1, false>},
    {{CLS_TREE, ENTRY_ENV, F_SPLIT, EU, 0, SRC_TABLE, 0}, false, l_tree_split<0, true>},
    {{CLS_TREE, ENTRY_ENV, F_SPLIT, RK, 0, SRC_TABLE, 0}, false, l_tree_split<1, true>},
    {{CLS_TREE, ENTRY_STEP, F_SPLIT, EU, 0, SRC_JIT, 0}, false, l_tree_split_jit<false>},
    {{CLS_TREE, ENTRY_STEP, F_SPLIT, RK, 0, SRC_JIT, 0}, false, l_tree_split_jit<false>},
    {{CLS_TREE, ENTRY_ENV, F_SPLIT, EU, 0, SRC_JIT, 0}, false, l_tree_split_jit<true>},
    {{CLS_TREE, ENTRY_ENV, F_SPLIT, RK, 0, SRC_JIT, 0}, false, l_tree_split_jit<true>},
    {{CLS_TREE, ENTRY_STEP, F_SPLIT2, EU, 0, SRC_TABLE, 0}, false, l_tree_split2<0, false>},
    {{CLS_TREE, ENTRY_STEP, F_SPLIT2, RK, 0, SRC_TABLE, 0}, false, l_tree_split2<1, false>},
    {{CLS_TREE, ENTRY_ENV, F_SPLIT2, EU, 0, SRC_TABLE, 0}, false, l_tree_split2<0, true>},
    {{CLS_TREE, ENTRY_ENV, F_SPLIT2, RK, 0, SRC_TABLE, 0}, false, l_tree_split2<1, true>},
    {{CLS_TREE, ENTRY_STEP, F_SPLIT2, EU, 0, SRC_JIT, 0}, false, l_tree_split2_jit<false>},
    {{CLS_TREE, ENTRY_STEP, F_SPLIT2, RK, 0, SRC_JIT, 0}, false, l_tree_split2_jit<false>},
    {{CLS_TREE, ENTRY_ENV, F_SPLIT2, EU, 0, SRC_JIT, 0}, false, l_tree_split2_jit<true>},
    {{CLS_TREE, ENTRY_ENV, F_SPLIT2, RK, 0, SRC_JIT, 0}, false, l_tree_split2_jit<true>},
};
constexpr int N_ROWS = int(sizeof(TABLE) / sizeof(TABLE[0]));

inline const Row *find_row(const Key &k) {
    for (const Row &r : TABLE) if (r.key == k) return &r;
    return nullptr;
}

// ---------------------------------------------------------------------------------------------------- RB_KERNEL_AUTO as data
// needs: what the handle must offer for a rule to apply
enum : int { NEED_MIRROR = RB_NEED_MIRROR, NEED_NO_MIRROR = RB_NEED_NO_MIRROR, NEED_SPLIT_TABLE = RB_NEED_SPLIT_TABLE, NEED_SPLIT2_TABLE = RB_NEED_SPLIT2_TABLE,
             NEED_LANE = RB_NEED_LANE };
constexpr long ANY_N = long(1) << 40;
constexpr int ANY = -1;
// measured crossovers (us per step; the sweeps are committed):
//   ball joints, plain step (profiles/r4_a/mid_sweep.log; eight lanes / two lanes / one lane per env): RK4 8 192 envs 3.39 / 4.11 / 5.07,
//     16 384 envs 4.63 / 4.23 / 5.11, 32 768 envs 7.25 / 4.50 / 5.23, 49 152 envs 9.84 / 6.19 / 5.35; Euler 4 096 envs 2.21 / 2.22 / 2.43,
//     8 192 envs 2.34 / 2.25 / 2.45, 16 384 envs 2.79 / 2.38 / 2.49, 32 768 envs 3.58 / 2.68 / 2.62; without a mirror plane
//     (profiles/r1_b/sweep.log): Euler 2.9 vs 3.5 us at 8 192 and a tie at 16 384, RK4 6.2 vs 7.0 at 16 384 and 10.2 vs 7.1 at 32 768
//   ball joints, fused env step (profiles/r5_a/env_octets_sweep.log, env_pairs_sweep.log): RK4 256 envs 3.38 / 4.39 / 5.36, 8 192 envs
//     3.81 / 4.59 / 5.55, 12 288 envs 4.97 / 4.65 / 5.63, 32 768 envs - / 5.17 / 6.05, 49 152 envs - / 7.02 / 6.44; Euler 8 192 envs
//     2.74 / 2.88 / 2.96, 12 288 envs 3.09 / 2.92 / 3.01, 32 768 envs - / 3.46 / 3.43
//   joint trees (profiles/r3_a, profiles/r5_a/split2_sweep.log): the five-wave split form while one workgroup per CU covers the batch,
//     the lean two-part form up to a wave on every SIMD (upper body 32 768 envs Euler 15.6 -> 13.35 us), one wave per 64 envs above
const rb_auto_rule AUTO_RULES[] = {
    // class     entry        integ needs              n >                      n <=                               form
    {CLS_BALL8, ENTRY_STEP, EU, NEED_MIRROR, 0, RB_TENDON_LANE_BATCH_PAIR_EULER, F_OCTET},
    {CLS_BALL8, ENTRY_STEP, EU, NEED_MIRROR, 0, RB_PAIR_BATCH_EULER, F_PAIR},
    {CLS_BALL8, ENTRY_STEP, RK, NEED_MIRROR, 0, RB_TENDON_LANE_BATCH_PAIR_RK4, F_OCTET},
    {CLS_BALL8, ENTRY_STEP, RK, NEED_MIRROR, 0, RB_PAIR_BATCH_RK4, F_PAIR},
    {CLS_BALL8, ENTRY_STEP, EU, NEED_NO_MIRROR, 0, RB_TENDON_LANE_BATCH_EULER, F_OCTET},
    {CLS_BALL8, ENTRY_STEP, RK, NEED_NO_MIRROR, 0, RB_TENDON_LANE_BATCH_RK4, F_OCTET},
    {CLS_BALL8, ENTRY_STEP, ANY, 0, 0, ANY_N, F_LANE},
    {CLS_BALL8, ENTRY_ENV, EU, 0, 0, RB_OCTET_ENV_BATCH_EULER, F_OCTET},
    {CLS_BALL8, ENTRY_ENV, RK, 0, 0, RB_OCTET_ENV_BATCH_RK4, F_OCTET},
    {CLS_BALL8, ENTRY_ENV, EU, NEED_MIRROR, 0, RB_PAIR_ENV_BATCH_EULER, F_PAIR},
    {CLS_BALL8, ENTRY_ENV, RK, NEED_MIRROR, 0, RB_PAIR_ENV_BATCH_RK4, F_PAIR},
    {CLS_BALL8, ENTRY_ENV, ANY, 0, 0, ANY_N, F_LANE},
    {CLS_BALL8, ENTRY_FUSED, ANY, 0, 0, ANY_N, F_LANE},
    {CLS_BALLX, ANY, ANY, 0, 0, ANY_N, F_LANE},
    {CLS_TREE, ANY, ANY, NEED_SPLIT_TABLE, 0, RB_TREE_SPLIT_BATCH, F_SPLIT},
    {CLS_TREE, ANY, ANY, NEED_SPLIT2_TABLE, RB_TREE_SPLIT_BATCH, RB_TREE_SPLIT2_BATCH, F_SPLIT2},
    {CLS_TREE, ANY, ANY, NEED_LANE, 0, ANY_N, F_LANE},
    {CLS_TREE, ANY, ANY, 0, 0, ANY_N, F_WAVE},
};
constexpr int N_AUTO_RULES = int(sizeof(AUTO_RULES) / sizeof(AUTO_RULES[0]));

inline int robot_class(const rb_sim *s) { return s->tree ? CLS_TREE : (s->ntx ? CLS_BALLX : CLS_BALL8); }
inline int abilities(const rb_sim *s) {
    int a = 0;
    if (!s->tree && !s->ntx) a |= s->pair_ok ? NEED_MIRROR : NEED_NO_MIRROR;
    if (s->tree && s->split_ok && s->split_baked) a |= NEED_SPLIT_TABLE;
    if (s->tree && s->split2_ok && s->split2_baked) a |= NEED_SPLIT2_TABLE;
    if (s->tree && tree_wants_lane_auto(s)) a |= NEED_LANE;
    return a;
}
// RB_KERNEL_AUTO: the form of the first rule that applies
inline int auto_form(const rb_sim *s, int entry) {
    const int cls = robot_class(s), have = abilities(s), integ = s->integrator == RB_EULER ? EU : RK;
    for (const rb_auto_rule &r : AUTO_RULES) {
        if (r.robot_class != cls || (r.entry != ANY && r.entry != entry) || (r.integrator != ANY && r.integrator != integ)) continue;
        if ((r.needs & have) != r.needs) continue;
        if (s->n > r.min_envs_exclusive && s->n <= r.max_envs) return r.kernel;
    }
    return s->tree ? F_WAVE : F_LANE;
}

// ---------------------------------------------------------------------------------------------------- resolution
// The form a launch of `entry` takes: the handle's pinned choice (rb_select_kernel) or AUTO's.  Joint trees degrade where a
// run-time-built kernel is not available (split -> one wave per 64 envs -> octets); build = false never starts a build (queries,
// graph-cache keys), build = true builds what the form needs outside stream captures - what the launch path has always done.
inline int resolve_form(rb_sim *s, int entry, bool build, std::string *why) {
    const int kind = entry == ENTRY_ENV ? 1 : 0;
    if (!s->tree) {
        if (s->ntx || entry == ENTRY_FUSED) return F_LANE;
        if (entry == ENTRY_STEP) return s->kernel;                       // rb_select_kernel keeps it: the choice, or AUTO's for the plain step
        return s->kernel_choice != RB_KERNEL_AUTO ? s->kernel_choice : auto_form(s, ENTRY_ENV);
    }
    int form = s->kernel_choice != RB_KERNEL_AUTO ? s->kernel_choice : auto_form(s, entry);
    if (form == F_SPLIT2) {
        if (s->split2_baked) return F_SPLIT2;
        rblj::Kernel &k = kind ? s->split2_env_k : s->split2_step_k;
        if (k.state == 1 || (build && build_split2_kernel(s, kind))) return F_SPLIT2;
        if (why) *why = "lean split kernel not available: " + k.why;
        return -1;                                                       // an explicit choice that cannot be served is an error, not a fallback
    }
    if (form == F_SPLIT) {
        if (s->split_baked) return F_SPLIT;
        rblj::Kernel &k = kind ? s->split_env_k : s->split_step_k;
        // (the plain step's kernel is built by rb_select_kernel; the env step's by rb_env_configure or here)
        if (k.state == 1 || (build && kind == 1 && build_split_kernel(s, 1))) return F_SPLIT;
        form = tree_wants_lane(s) ? F_LANE : F_WAVE;
    }
    if (form == F_LANE) {
        if (s->lane_baked) return F_LANE;
        if (tree_wants_lane(s) && (build ? lane_kernel(s, kind) : (kind ? &s->lane_env_k : &s->lane_step_k))->state == 1) return F_LANE;
        return F_WAVE;
    }
    return F_WAVE;
}

inline bool resolve(rb_sim *s, int entry, bool build, Key &k, std::string *why = nullptr) {
    const int form = resolve_form(s, entry, build, why);
    if (form < 0) return false;
    k = Key{robot_class(s), entry, form, s->integrator == RB_EULER ? EU : RK, 0, SRC_KERNARG, 0};
    if (s->tree) {
        if (form == F_WAVE) { k.variant = s->tree_host.dev.single_pass != 0 ? 1 : 0; return true; }
        const bool table = form == F_LANE ? s->lane_baked : (form == F_SPLIT ? s->split_baked : s->split2_baked);
        k.src = table ? SRC_TABLE : SRC_JIT;
        k.block = form == F_LANE ? 64 : 0;
        return true;
    }
    if (form == F_OCTET) { k.block = 64; return true; }
    if (form == F_PAIR) {
        k.block = s->n <= RB_PAIR_SMALL_BATCH ? 64 : 256;
        k.src = s->pair_baked ? SRC_TABLE : SRC_KERNARG;
        k.variant = s->pair_mirror;
        return true;
    }
    k.block = s->n <= RB_SMALL_BATCH ? 64 : 256;
    if (s->ntx) return true;
    if (build) maybe_jit(s);                                             // (large batches of another 8-tendon robot: its own instances, once)
    if (s->baked) k.src = SRC_TABLE;
    else if (k.block == 256 && s->jit_state == 1) k.src = SRC_JIT;       // (hiprtc instances exist for the large-batch configuration only)
    return true;
}

// the row a launch of `entry` takes on this handle (nullptr + message: no kernel for the request)
inline const Row *row_for(rb_sim *s, int entry, bool build, std::string *why = nullptr) {
    Key k;
    if (!resolve(s, entry, build, k, why)) return nullptr;
    const Row *r = find_row(k);
    if (!r && why) *why = "no kernel instance for this robot / form / batch (dispatch table)";
    return r;
}

// every launch of the library's three entry kinds
inline int dispatch(rb_sim *s, int entry, const Launch &L) {
    std::string why;
    const Row *r = row_for(s, entry, true, &why);
    if (!r) return fail(RB_EUNSUPPORTED, why);
    if (!r->ranges && !(L.i0 == 0 && L.cnt == s->n)) return fail(RB_EUNSUPPORTED, "this kernel form steps whole batches only (rb_range_capable)");
    if (s->tree) s->kernel = r->key.form;                                // what rb_info reports for joint trees: the form last launched
    const int rc = r->launch(s, L);
    if (rc) return rc;
    RB_HIP(hipGetLastError());
    return RB_OK;
}
