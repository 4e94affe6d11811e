/* roboy_policy.h - C ABI of the fused MLP policy step for the PPO consumer.
 *
 * NOT part of the drop-in boundary of the simulation path (that is roboy_sim.h): the reference's consumer is
 * stable_baselines' PPO2("MlpPolicy", env) (/root/reference/gym_roboy/train_parallel.py:28-35), whose policy is two
 * tanh layers of 64 units for the action mean and two for the value, with a state-independent log-std
 * (gym_roboy_amd/ppo.py: MlpPolicy restates it on torch).  With the env step at ~12 us for 262 144 envs, the torch
 * forward pass + sampling of that policy (about 30 small kernels, 370 us) is what a rollout step costs; this
 * library evaluates  obs -> (action sample, log-probability, value)  in ONE kernel on the matrix cores
 * (v_mfma_f32_32x32x2_f32: exact f32, a k-ordered fmaf chain), gym_roboy_amd/csrc/mlp_policy.hip.
 *
 * Plain C, pointers and sizes only; device pointers are HIP device memory of the current device; `stream` is a
 * hipStream_t (0 = the default stream).  Every function returns 0 on success, a negative code otherwise
 * (rp_last_error() has the message).
 */
#ifndef ROBOY_POLICY_H
#define ROBOY_POLICY_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define RP_ABI_VERSION 1
#define RP_HIDDEN 64          /* units per hidden layer (stable_baselines' MlpPolicy) */
#define RP_MAX_OBS 95         /* obs_dim + 1 (bias column) <= 96 */
#define RP_MAX_ACT 64

enum { RP_OK = 0, RP_EINVAL = -1, RP_EHIP = -2, RP_EUNSUPPORTED = -3 };

/* parameters of one MlpPolicy in torch's layout: Linear.weight is [out][in] row-major, bias [out] */
typedef struct rp_mlp_params {
    const float *pi_w1, *pi_b1, *pi_w2, *pi_b2, *pi_w3, *pi_b3;   /* [64][obs], [64], [64][64], [64], [act][64], [act] */
    const float *vf_w1, *vf_b1, *vf_w2, *vf_b2, *vf_w3, *vf_b3;   /* [64][obs], [64], [64][64], [64], [1][64],   [1]   */
    const float *log_std;                                          /* [act] */
} rp_mlp_params;

int rp_abi_version(void);
const char *rp_last_error(void);

/* Number of floats of the packed parameter blob for these dimensions (the matrix-core operand order: each weight
 * sits where the lane that feeds it to the MFMA reads it), or a negative code if the dimensions are not supported. */
int64_t rp_packed_floats(int obs_dim, int act_dim);

/* Host: reorder the parameters (host pointers) into the packed blob (host, rp_packed_floats() floats).  The order
 * depends on the dimensions only, so a caller may pack an index-valued parameter set once and use the result as a
 * gather map on the device. */
int rp_pack(const rp_mlp_params *host_params, int obs_dim, int act_dim, float *packed_host);

/* One policy step for n observations (device pointers): obs [n][obs_dim] -> act [n][act_dim] = mean + std * eps,
 * logp [n] = log N(act; mean, std), value [n]; mean [n][act_dim] is written too unless NULL.  eps: standard normal
 * from Philox4x32-10 keyed (seed; sample index + sample_offset, step) by Box-Muller - the same draw whatever the
 * batch is sharded into; d_step_base (device, may be NULL): added to `step` when the kernel runs, so that a
 * captured launch (hipGraph replay) draws fresh noise on every replay; deterministic != 0: act = mean.
 * Asynchronous on `stream`. */
int rp_act_dev(const float *d_packed, const float *d_obs, float *d_act, float *d_logp, float *d_value, float *d_mean,
               int64_t n, int obs_dim, int act_dim, uint64_t seed, uint64_t sample_offset, uint32_t step,
               const uint32_t *d_step_base, int deterministic, void *stream);

#ifdef __cplusplus
}
#endif
#endif
