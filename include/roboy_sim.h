/* roboy_sim.h - C ABI of the MI355X batched tendon-robot physics step.
 *
 * Drop-in boundary for gym-roboy's SimulationClient plugin interface
 * (reference: gym_roboy/envs/simulations/simulation_client.py:6-23).  In the
 * reference every method of that interface is one ROS service round-trip into
 * the external CARDSflow simulator (ros_simulation_client.py:32-81); here each
 * is one in-process call that advances / reads N independent environments
 * whose state lives in HBM.  Plain pointers and sizes only; no torch types.
 *
 *   reference method (file:line)                         entry point here
 *   ---------------------------------------------------  ----------------------
 *   RosSimulationClient.__init__        (ros_..py:16-30)  rb_create
 *   forward_step_command(action)        (ros_..py:48-60)  rb_step / rb_step_dev
 *   forward_reset_command()             (ros_..py:32-38)  rb_reset
 *   read_state()                        (ros_..py:66-71)  rb_read_state
 *   get_new_goal_joint_angles()         (ros_..py:73-81)  rb_sample_goals
 *   _check_service_available_or_timeout (ros_..py:62-64)  return codes + rb_last_error
 *   RoboyEnv.step env layer             (roboy_env.py:51-70,92-134)  rb_env_step_dev
 *
 * Layouts.  Host-facing arrays are row-major [n_envs][n] float32 (what numpy
 * and a policy network produce).  Device state is struct-of-arrays: joint
 * angle plane j is q[j*n_envs .. (j+1)*n_envs), same for velocities, plus one
 * uint32 feasibility word per env.  Device action slabs are row-major
 * [n_envs][n_t] float32 (one 32-byte record per MsjRobot env: two 16-byte
 * loads per lane, contiguous across the wave).
 *
 * Threading: a handle owns one device and one stream; calls on one handle
 * must be serialised by the caller; handles are independent (one process per
 * GPU for the multi-GPU configuration).  All functions return RB_OK (0) or an
 * error code, never abort; rb_last_error() gives the message (thread-local).
 */
#ifndef ROBOY_SIM_H
#define ROBOY_SIM_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RB_ABI_VERSION 5

enum rb_status {
    RB_OK = 0,
    RB_EINVAL = 1,       /* bad argument / malformed robot description      */
    RB_EUNSUPPORTED = 2, /* robot structure has no HIP kernel (yet)         */
    RB_EHIP = 3,         /* HIP runtime error (no device, launch failure)   */
    RB_ENOMEM = 4
};

enum rb_integrator { RB_EULER = 0, RB_RK4 = 1 };

/* kernel variants (rb_select_kernel): */
enum rb_kernel {
    RB_KERNEL_AUTO = 0,
    RB_KERNEL_ENV_PER_LANE = 1,    /* one env per lane: throughput form        */
    RB_KERNEL_TENDON_PER_LANE = 2, /* 8 lanes per env + DPP reductions: latency form */
    RB_KERNEL_ENV_PER_WAVE = 3,    /* generic joint-tree robots: a few envs per wave, eight lanes per link, LDS */
    RB_KERNEL_ENV_PER_LANE_SPLIT = 4,/* joint trees, small batches: one env per lane, but several waves per group of 64
                                      envs - one per set of branches of the tree - so that a step waits for a part of
                                      the instruction stream only */
    RB_KERNEL_ENV_PER_LANE_SPLIT2 = 6,/* joint trees, batches between "one workgroup of form 4 per CU" and "a wave on every SIMD"
                                      (the upper body: 16 384 < n <= 32 768 envs): the same code in TWO part waves per 64 envs
                                      with a workgroup lean enough in LDS for two per CU; the library's own choice in that
                                      range for the committed upper body, built by hiprtc on request for other robots */
    RB_KERNEL_LANE_PAIR = 5          /* ball-joint robots with a mirror plane (MsjRobot): two lanes per env - the odd lane
                                      steps the env's mirror image with the same code and constants, four tendons each,
                                      torque sums swapped by DPP: twice the waves of the env-per-lane form for the same
                                      batch, a per-wave instruction chain 0.64x as long */
};

/* Robot description, format "roboy-tendon-robot/1" (Python
 * mirror: gym_roboy_amd/envs/robots/description.py).  All arrays are owned by
 * the caller and only read during rb_create. */
typedef struct rb_robot_desc {
    int32_t n_q;             /* joints = moving links                         */
    int32_t n_t;             /* tendons                                       */
    int32_t n_vp;            /* total via-points                              */
    int32_t _pad;
    const int32_t *parent;   /* [n_q] parent link, -1 = base                  */
    const double *axis;      /* [n_q][3] unit joint axis, parent frame        */
    const double *origin;    /* [n_q][3] joint origin, parent frame           */
    const double *mass;      /* [n_q]                                         */
    const double *com;       /* [n_q][3] centre of mass, link frame           */
    const double *inertia;   /* [n_q][6] xx,yy,zz,xy,xz,yz about COM          */
    const double *armature;  /* [n_q] reflected actuator inertia              */
    const double *damping;   /* [n_q] viscous joint damping                   */
    const double *q_lo;      /* [n_q] feasible region, lower                  */
    const double *q_hi;      /* [n_q] feasible region, upper                  */
    const double *qd_max;    /* [n_q] joint speed limit                       */
    double gravity[3];
    const int32_t *vp_offset;/* [n_t+1] CSR offsets into the via-point arrays */
    const int32_t *vp_link;  /* [n_vp] link of each via-point, -1 = base      */
    const double *vp_pos;    /* [n_vp][3] position in that link's frame       */
    const double *f_max;     /* [n_t] maximum isometric muscle force          */
    double kp, setpoint_scale, v_max, fl_width, kpe, e0, fv_a, fv_n;
} rb_robot_desc;

typedef struct rb_sim rb_sim;

typedef struct rb_sim_info {
    int64_t n_envs;
    int32_t n_q, n_t;
    int32_t integrator, n_substeps;
    int32_t kernel;             /* rb_kernel actually in use                  */
    int32_t device;
    double step_size;
    int64_t bytes_per_env_step; /* algorithmic HBM bytes: 4*(4 n_q + n_t + 1) */
    int64_t env_id_offset;
} rb_sim_info;

/* env-layer configuration for rb_env_* (reference RoboyEnv.__init__ kwargs and
 * constants, roboy_env.py:12-28) */
typedef struct rb_env_config {
    int32_t joint_vel_penalty;      /* roboy_env.py:13                        */
    int32_t goal_bonus;             /* is_agent_getting_bonus_for_reaching_goal */
    int32_t max_episode_length;     /* 400, roboy_env.py:28                   */
    int32_t auto_reset;             /* 1: on done also reset the env, as the
                                       reference's SubprocVecEnv workers do
                                       (train_parallel.py:29); 0: RoboyEnv.step
                                       alone (goal resampled only, :67-68)    */
    float penalty_boundary;         /* 1,    roboy_env.py:26                  */
    float bonus_goal;               /* 1000, roboy_env.py:27                  */
    float angle_lo, angle_hi;       /* joint-angle box (msj_robot.py:9)       */
    float vel_lo, vel_hi;           /* joint-velocity box (msj_robot.py:10)   */
    float action_lo, action_hi;     /* tendon set-point box (msj_robot.py:16) */
    float goal_angle_tol;           /* _MAX_DISTANCE_JOINT_ANGLE / 200 (:127) */
    float goal_vel_tol;             /* _MAX_DISTANCE_JOINT_VELS / 5    (:130) */
} rb_env_config;

const char *rb_last_error(void);
int rb_abi_version(void);
int rb_device_count(int *count);

/* rb_create: build a batch of n_envs environments on `device`.
 * env_id_offset shards one logical batch over several handles/GPUs: all
 * counter-based random streams are keyed by (seed, env_id_offset + i), so
 * results do not depend on how the batch is split.
 * The robot decides the kernels: one body on an x-y-z ball joint with 8 tendons
 * (MsjRobot) gets the specialised closed-form kernels; the same class with 1..16
 * tendons the closed form with a run-time tendon count (env-per-lane only); any
 * other joint tree (up to 32 joints / 64 tendons) the joint-tree kernels (articulated-body
 * algorithm): one env per lane running code generated for the robot (the committed upper
 * body ahead of time - with a several-waves-per-env-group form for small batches - any
 * other robot through hiprtc) or, without such code, octets of lanes per link. */
int rb_create(const rb_robot_desc *robot, int64_t n_envs, int integrator,
              double step_size, int n_substeps, int device, uint64_t seed,
              int64_t env_id_offset, rb_sim **out);
void rb_destroy(rb_sim *sim);
int rb_info(const rb_sim *sim, rb_sim_info *info);
/* rb_select_kernel: pin the kernel form (RB_KERNEL_*), or hand the choice back to the library (RB_KERNEL_AUTO, the default).
 * What is bit-equal to what: every form evaluates the same model (each passes the same parity tests against the fp64 oracle,
 * 2e-5), but the forms order their floating-point sums differently, so results are BIT-identical only within one form -
 *   - across batch splits (env_id_offset shards, sub-ranges, rollout chains): bit-identical as long as every piece runs the
 *     same form.  RB_KERNEL_AUTO chooses by the HANDLE's batch size (ball joints with a mirror plane: eight lanes per env up to
 *     4 096 / 12 288 envs for Euler / RK4, two lanes per env up to 16 384 / 32 768, one env per lane above), so shards on either
 *     side of a threshold agree to ~1 ulp per step, not bit for bit: pin the form (the same rb_select_kernel on every
 *     handle) where bit equality across shard sizes is wanted;
 *   - rb_env_step_dev against rb_step_dev + the reference's env arithmetic: bit-identical states when the handle's form is
 *     pinned to 1, 2 or 5.  Under RB_KERNEL_AUTO the env layer chooses by its own thresholds: eight lanes per env up to 8 192 envs,
 *     two lanes per env up to 24 576 / 32 768 envs (robots with a mirror plane), one env per lane otherwise;
 *   - the env-per-lane form itself has two instances by the handle's batch size: up to 65 536 envs (64-thread workgroups,
 *     tendon loop written out) and above (256-thread workgroups; RK4: the stages as a rolled loop over running sums) - equal to
 *     ~1 ulp, bit-identical only among handles on the same side of 65 536 envs. */
int rb_select_kernel(rb_sim *sim, int kernel);
/* How the env-per-lane kernels of this handle get the robot's constants: RB_SPEC_NONE = through the kernarg
 * (scalar loads); RB_SPEC_TABLE = instances compiled ahead of time on the reference's MsjRobot (the handle's
 * constants equal that table bit for bit); RB_SPEC_JIT = instances compiled with hiprtc on this robot's own
 * constants (8-tendon ball-joint robots, batches above 65 536 envs; built by the first call that needs them, or
 * by this one; ROBOY_SIM_JIT=0 disables it).  All three are the same source and pass the same parity tests;
 * the specialised ones run ~6 % faster.  Returns -1 for a null handle; after RB_SPEC_NONE rb_last_error() says why. */
enum { RB_SPEC_NONE = 0, RB_SPEC_TABLE = 1, RB_SPEC_JIT = 2 };
int rb_specialization(rb_sim *sim);
/* The code objects hiprtc builds are kept on disk (ROBOY_SIM_JIT_CACHE = a directory; default $XDG_CACHE_HOME/gym_roboy_amd or
 * ~/.cache/gym_roboy_amd; "0" = no cache), keyed by source text (robot constants included), architecture and library build:
 * a second process on the same robot loads in milliseconds what took hiprtc 1-80 s.  Process-wide counts since load:
 * hits = builds served from the cache, compiles = hiprtc compilations, stores = files written.  Any pointer may be NULL. */
void rb_jit_cache_stats(int64_t *hits, int64_t *compiles, int64_t *stores);
/* The stream every launch, copy and synchronisation of this handle uses.  NULL = the
 * handle's own (non-blocking) stream; RB_STREAM_DEVICE_DEFAULT = the device's default
 * (null) stream, which is what a framework's "current stream" is unless the caller made
 * another one current (its handle is 0 and cannot be told from NULL, hence the
 * sentinel); anything else = that hipStream_t.  Drains the previous stream first. */
#define RB_STREAM_DEVICE_DEFAULT ((void *)(intptr_t)-1)
int rb_set_stream(rb_sim *sim, void *hip_stream);
int rb_synchronize(rb_sim *sim);

/* ---- host-buffer entry points (synchronous; plumbing, not the hot loop) ---- */
int rb_reset(rb_sim *sim, const uint8_t *mask /* [n_envs] or NULL = all */);
int rb_set_state(rb_sim *sim, const float *q, const float *qd,
                 const uint8_t *feasible /* NULL = all feasible */);
int rb_read_state(rb_sim *sim, float *q, float *qd, uint8_t *feasible);
/* set-point = act_scale * act; act_scale = 1 when the caller has already
 * rescaled to the robot's set-point box (roboy_env.py:54-57) */
int rb_step(rb_sim *sim, const float *act, float act_scale,
            float *q, float *qd, uint8_t *feasible);
int rb_sample_goals(rb_sim *sim, const uint8_t *mask, float *goal_q);

/* ---- device-pointer entry points (asynchronous on the handle's stream) ---- */
/* The library's own state buffers, in place.  Layout by kernel class: ball-joint robots keep SoA planes
 * q[n_q][n_envs], qd[n_q][n_envs] (one env per lane reads a plane element each); joint-tree robots keep
 * env-major rows q[n_envs][n_q], qd[n_envs][n_q] (a wave owns a few whole envs).  feasible[n_envs] either way. */
int rb_state_ptrs(rb_sim *sim, float **d_q, float **d_qd, uint32_t **d_feasible);
int rb_step_dev(rb_sim *sim, const float *d_act, float act_scale);
/* n_steps steps, one kernel boundary per step; step t reads slab (t % ring) of d_act_ring
 * ([ring][n_envs][n_t]).  use_graph != 0 replays the launches from captured hipGraphs -
 * for large batches as rb_rollout_chains() independent chains, i.e. one launch per step
 * and HALF of the batch (below); use_graph = 0 launches once per step over the whole batch. */
int rb_rollout_dev(rb_sim *sim, const float *d_act_ring, int ring, int n_steps,
                   float act_scale, int use_graph);
/* How rb_rollout_dev (use_graph = 1) steps this handle: 1 = one launch over the whole batch per step; 2 = the two halves of the batch
 * as two independent chains of launches, each a linear graph on its own stream (large batches - ball joints from 98 304 envs RK4 /
 * 262 144 envs Euler, the one-wave joint-tree form from 32 768 / 65 536: one half's launch gaps and load / store phases lie under the
 * other half's arithmetic - MsjRobot RK4 at 262 144 envs 16.6 -> 13.0 us per step; envs are independent, the results are the same
 * bit for bit; ROBOY_SIM_CHAINS=1 switches it off, 2-4 force a count).  Eager rollouts (use_graph = 0) and rb_step_dev launch once. */
int rb_rollout_chains(rb_sim *sim);
/* 0 = the library's choice (above), 1..4 = that many chains for the kernel forms that can be stepped in sub-ranges (others keep 1):
 * what bench.py uses to time the one-launch-per-step form beside the default one. */
int rb_set_rollout_chains(rb_sim *sim, int chains);
/* Open-loop rollout fused into ONE launch: every env advances n_steps steps, the
 * state stays in registers in between and only the action of each step is read
 * (slab t % ring of d_act_ring).  For action sequences that are known up front
 * (random-action rollouts, sampling-based planning); NOT the per-step contract of
 * rb_step_dev / rb_rollout_dev (no policy can sit between steps), so bench.py
 * reports it separately and never as the headline.  Bit-identical to n_steps
 * calls of rb_step_dev with the env-per-lane kernel form selected (the
 * tendon-per-lane form RB_KERNEL_AUTO picks for small batches sums the tendon
 * torques in another order: same result to ~1e-6).  Ball-joint robots only. */
int rb_rollout_fused_dev(rb_sim *sim, const float *d_act_ring, int ring, int n_steps, float act_scale);
/* synthetic i.i.d. U[-1,1) actions: Philox4x32-10, key = seed,
 * counter = (env id, step, stream 0) */
int rb_fill_actions_dev(rb_sim *sim, float *d_act, uint32_t step);
int rb_sample_goals_dev(rb_sim *sim, const uint8_t *d_mask, float *d_goal_q /* [n_q][n_envs] */);

/* ---- fused env layer (next row of the scope table) ---- */
int rb_env_configure(rb_sim *sim, const rb_env_config *cfg);
int rb_env_reset_dev(rb_sim *sim, float *d_obs /* [n_envs][3 n_q] */);
/* Host-buffer entry (synchronous): overwrite every env's goal ([n_envs][n_q]) and,
 * if step_num is not NULL, its episode step counter - what assigning
 * RoboyEnv._goal_state / .step_num does in the reference's own tests
 * (gym_roboy/envs/tests/test_roboy_env.py:62-66,172-176); lets a caller replay recorded
 * (state, goal, counter) triples through rb_env_step_dev. */
int rb_env_set_goal(rb_sim *sim, const float *goal_q, const uint32_t *step_num /* [n_envs] or NULL */);
int rb_env_step_dev(rb_sim *sim, const float *d_act /* [n_envs][n_t] in [-1,1] */,
                    float *d_obs, float *d_reward, uint32_t *d_done);
/* ---- sub-ranges: envs [first_env, first_env + n_envs) on a stream of the caller's choice ----
 * The per-step entry points above launch ONCE per step over the whole batch - which, for a large batch, leaves the launch
 * gap and the load / store phases of a single generation of waves uncovered (rb_rollout_dev's chains exist for that).  A
 * closed-loop caller gets the same overlap by running its loop as two independent chains over the two halves of the batch
 * on two streams - its policy kernel and this step per half (gym_roboy_amd/ppo.py does: the reference's consumer,
 * train_parallel.py:28-35) - forked and joined by the caller, once per rollout.  (Forking and joining inside every step
 * call does not pay: a join costs ~8 us on this stack, profiles/r4_a/region_timeline.log.)  Envs are independent, so
 * disjoint ranges may run concurrently and the results do not depend on the split.  The array arguments are those of the
 * WHOLE batch (the library applies the offsets); first_env is a multiple of 256; hip_stream NULL = the handle's stream.
 * rb_range_capable(): bit 0 = rb_step_range_dev takes sub-ranges with the handle's kernel form (ball joints with 8 tendons
 * except the tendon-per-lane form; joint trees in the one-wave-per-64-envs form), bit 1 = rb_env_step_range_dev does (ball
 * joints always; joint trees in that form); a clear bit = whole batches only: a WHOLE batch (first_env = 0, n_envs = all) is taken
 * by every form on the caller's stream, a true sub-range is refused with RB_EUNSUPPORTED.  (The `ranges` field of the dispatch
 * table's rows, below, says the same per kernel instance.)
 * Lifetime: the handle remembers every distinct caller stream it has launched a range on (an event behind the last launch;
 * 8 entries, least recently used reused after a wait), and rb_destroy / rb_select_kernel / rb_set_stream wait for that work
 * like for the handle's own streams - the caller need not join before closing.  The caller's stream may already be
 * destroyed by then (the event is waited for, not the stream).  During a stream CAPTURE of the caller's stream nothing is
 * in flight and nothing is remembered: run-time kernel builds are deferred (a form that has no kernel yet is refused with
 * RB_EUNSUPPORTED rather than built inside the capture), and whoever replays the captured graph keeps the handle alive and
 * its kernel form unchanged while replays are in flight. */
int rb_range_capable(rb_sim *sim);
int rb_step_range_dev(rb_sim *sim, int64_t first_env, int64_t n_envs, void *hip_stream, const float *d_act, float act_scale);
int rb_env_step_range_dev(rb_sim *sim, int64_t first_env, int64_t n_envs, void *hip_stream,
                          const float *d_act, float *d_obs, float *d_reward, uint32_t *d_done);
/* episode statistics summed over this handle's envs since the last reset of
 * the accumulators: [sum episode return, sum return^2, n_episodes, sum episode
 * length, n_goal_reached, n_infeasible_env_steps, n_env_steps, sum reward]
 * (fp64).  Without rb_env_configure only [5] (envs flagged infeasible at the
 * time of the call) and [6] are non-zero.  The _dev form copies them into caller-owned device memory (e.g. a
 * torch tensor that is then all-reduced over RCCL); the host form synchronises. */
int rb_env_stats(rb_sim *sim, double *stats8, int reset);
int rb_env_stats_dev(rb_sim *sim, double *d_stats8, int reset);

/* ---- which kernel instance a call launches: the library's dispatch table, readable (ABI 5) ----
 * Every launch of the three entry kinds goes through ONE table of kernel instances keyed by (robot class, entry kind, kernel form,
 * integrator, workgroup size, constants source, variant); RB_KERNEL_AUTO's thresholds are a list of rules (first match wins).  Both
 * are static data of the library: readable without a GPU, so tests enumerate them instead of restating them. */
enum rb_robot_class { RB_CLASS_BALL8 = 0,   /* one body on an x-y-z ball joint, 8 tendons (MsjRobot's class)         */
                      RB_CLASS_BALLX = 1,   /* the same with 1..16 tendons (run-time count)                           */
                      RB_CLASS_TREE = 2 };  /* any other joint tree                                                   */
enum rb_entry_kind { RB_ENTRY_STEP = 0,           /* rb_step, rb_step_dev, rb_step_range_dev, rb_rollout_dev         */
                     RB_ENTRY_ENV_STEP = 1,       /* rb_env_step_dev, rb_env_step_range_dev                          */
                     RB_ENTRY_FUSED_ROLLOUT = 2 };/* rb_rollout_fused_dev                                            */
typedef struct rb_dispatch_row {
    int32_t robot_class;   /* rb_robot_class                                                                          */
    int32_t entry;         /* rb_entry_kind                                                                           */
    int32_t kernel;        /* an rb_kernel value, never RB_KERNEL_AUTO                                                      */
    int32_t integrator;    /* rb_integrator                                                                           */
    int32_t block;         /* threads per workgroup; 0 = decided by the robot (octet waves, split parts)              */
    int32_t constants;     /* RB_SPEC_NONE (kernarg) / RB_SPEC_TABLE (ahead-of-time instances) / RB_SPEC_JIT (hiprtc) */
    int32_t variant;       /* two lanes per env: mirror plane (0 x-z, 1 y-z); octet joint-tree kernels: single-pass tables; else 0 */
    int32_t ranges;        /* 1: takes sub-ranges of the batch (rb_*_range_dev, rollout chains); 0: whole batches only */
} rb_dispatch_row;
/* what a handle must offer for an AUTO rule to apply (rb_auto_rule.needs, a bit set) */
enum { RB_NEED_MIRROR = 1,        /* ball joints: the robot has a mirror plane (two-lanes-per-env form possible)       */
       RB_NEED_NO_MIRROR = 2,     /* ball joints: it has none                                                          */
       RB_NEED_SPLIT_TABLE = 4,   /* joint trees: ahead-of-time instances of the five-wave split form (the committed upper body) */
       RB_NEED_SPLIT2_TABLE = 8,  /* ... of the lean two-part split form                                               */
       RB_NEED_LANE = 16 };       /* joint trees: one-wave-per-64-envs kernels at hand (ahead of time) or worth building (hiprtc: batch
                                     >= 16 384 envs, generated code within the register file; ROBOY_SIM_JIT) */
typedef struct rb_auto_rule {
    int32_t robot_class;          /* rb_robot_class                                                                    */
    int32_t entry;                /* rb_entry_kind, -1 = any                                                           */
    int32_t integrator;           /* rb_integrator, -1 = any                                                           */
    int32_t needs;                /* RB_NEED_* bits that must all be set                                               */
    int64_t min_envs_exclusive;   /* applies to handles with min_envs_exclusive < n_envs <= max_envs                   */
    int64_t max_envs;
    int32_t kernel;               /* the rb_kernel AUTO picks                                                          */
    int32_t _pad;
} rb_auto_rule;
/* batch-size thresholds of the launch configuration that are not kernel forms */
typedef struct rb_launch_thresholds {
    int64_t small_batch;          /* ball joints, one env per lane: 64-thread workgroups up to here, 256 above (and hiprtc / baked large-batch instances) */
    int64_t pair_small_batch;     /* two lanes per env: the same switch                                                */
    int64_t chain_batch_rk4, chain_batch_euler;             /* rb_rollout_dev: two chains from here on (ball joints)  */
    int64_t chain_batch_tree_rk4, chain_batch_tree_euler;   /* ... joint trees in the one-wave form                   */
    int64_t eager_head_batch_rk4; /* rb_rollout_dev: one ring turn of plain launches in front of the graphs (ball joints, RK4) */
    int64_t tree_jit_batch;       /* joint trees without ahead-of-time instances: AUTO builds the one-wave kernels from here on */
} rb_launch_thresholds;
int rb_dispatch_rows(const rb_dispatch_row **rows);     /* returns the row count; *rows = the library's static table */
int rb_auto_rules(const rb_auto_rule **rules);          /* returns the rule count; first match wins               */
int rb_get_launch_thresholds(rb_launch_thresholds *out);
/* The row the NEXT launch of `entry` over the whole batch takes on this handle (RB_EUNSUPPORTED + message if none).  Builds the
 * run-time kernels that launch would build (hiprtc; never inside a stream capture - then the row of what is at hand). */
int rb_dispatch_current(rb_sim *sim, int entry, rb_dispatch_row *out);

/* device memory helpers so a ctypes caller needs no HIP binding of its own */
int rb_malloc(rb_sim *sim, int64_t bytes, void **d_ptr);
int rb_free(rb_sim *sim, void *d_ptr);
int rb_memcpy_h2d(rb_sim *sim, void *d_dst, const void *h_src, int64_t bytes);
int rb_memcpy_d2h(rb_sim *sim, void *h_dst, const void *d_src, int64_t bytes);

#ifdef __cplusplus
}
#endif
#endif /* ROBOY_SIM_H */
