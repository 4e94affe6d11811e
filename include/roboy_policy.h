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

#define RP_ABI_VERSION 2
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

/* ---- the PPO minibatch gradient (gym_roboy_amd/ppo.py: _minibatch_loss is the torch statement) ----
 * Loss = mean_i max(-A_i r_i, -A_i clip(r_i, 1 -+ c)) + vf_coef * mean_i 1/2 max((v_i - R_i)^2, (vclip_i - R_i)^2),
 * r_i = exp(logp(act_i | obs_i) - logp_old_i), vclip_i = v_old_i + clip(v_i - v_old_i, -+c); A_i is the caller's
 * (already normalised) advantage.  The entropy bonus of a state-independent log-std has the constant gradient
 * -ent_coef per log-std component and is left to the caller. */

/* Generalised advantage estimation over a rollout (device pointers, [n_steps][n_envs] row-major; done[t] = step t
 * ended an episode, as 0 / 1 floats; last_val [n_envs] = value of the observation after the last step):
 * delta_t = rew_t + gamma * V_{t+1} * (1 - done_t) - V_t,  adv_t = delta_t + gamma * lam * (1 - done_t) * adv_{t+1},
 * ret_t = adv_t + V_t.  One launch, one env per thread. */
int rp_gae_dev(const float *d_rew, const float *d_val, const float *d_done, const float *d_last_val, float gamma, float lam,
               float *d_adv, float *d_ret, int n_steps, int64_t n_envs, void *stream);

/* floats of the blob rp_pack_train() writes: the rp_pack() blob followed by the transposed weights; negative if the
 * gradient kernels do not support the dimensions: obs_dim <= 63, and the operands of the action net plus the per-wave
 * scratch must fit the 160 KB of LDS (60 -> 38 does, 63 -> 64 does not) */
int64_t rp_train_packed_floats(int obs_dim, int act_dim);
int rp_pack_train(const rp_mlp_params *host_params, int obs_dim, int act_dim, float *packed_host);

/* The gradient vector d_grad: two blocks of rp_grad_floats() / 2 floats, action net then value net, each in torch
 * layout  [w1 (64 x obs), b1 (64), w2 (64 x 64), b2 (64), w3 (out x 64), b3 (out), log_std (out; zero for the value
 * net), loss term, 3 spare]  padded to a multiple of 4 (out = act_dim / 1). */
int64_t rp_grad_floats(int obs_dim, int act_dim);
int64_t rp_ppo_workspace_floats(int obs_dim, int act_dim, int64_t batch);

/* Device pointers.  d_index == NULL: obs [batch][obs_dim], act [batch][act_dim], adv / logp_old / val_old / ret
 * [batch].  d_index != NULL (int64 [batch]): sample i of the minibatch is ROW d_index[i] of obs, act, logp_old,
 * val_old, ret (the whole rollout's tensors - no gathered copies).  The advantage: with d_adv_stats == NULL, adv
 * [batch] is in minibatch order and already normalised by the caller; with d_adv_stats = the {mean, 1 / (std + 1e-8)}
 * pair rp_adv_stats_dev wrote, adv is the rollout's raw advantage, indexed like the rest and normalised per sample
 * in the kernel.  d_grad: rp_grad_floats() floats (overwritten), d_workspace: rp_ppo_workspace_floats() floats.
 * Four launches on `stream` (one per net, two reductions); asynchronous.  Like every rp_*_dev entry point it runs
 * on the device its first device pointer lives on (made current for the call). */
int rp_ppo_grad_dev(const float *d_packed_train, const float *d_obs, const float *d_act, const float *d_adv,
                    const float *d_adv_stats, const float *d_logp_old, const float *d_val_old, const float *d_ret,
                    const int64_t *d_index, int64_t batch, int obs_dim, int act_dim, float cliprange, float vf_coef,
                    float *d_grad, float *d_workspace, void *stream);

/* ---- the rest of a PPO update (gym_roboy_amd/ppo.py: update) ---- */
/* The sample order of an epoch: out[j] = P(first + j), j < count, for a bijection P of [0, n) keyed by `key` (a
 * four-round Feistel network on the next even power of two, cycle-walked into [0, n)): what torch.randperm(n) is to
 * the torch path, evaluated per element instead of sorting n keys.  _host: the same function on the CPU. */
int rp_perm_dev(uint64_t key, int64_t n, int64_t first, int64_t count, int64_t *d_out, void *stream);
int rp_perm_host(uint64_t key, int64_t n, int64_t first, int64_t count, int64_t *out);
/* d_stats2 = {mean, 1 / (std + 1e-8)} (std: the unbiased estimate, as torch.Tensor.std) of adv[d_index[i]], i < batch
 * (d_index == NULL: adv[i]).  d_scratch: rp_adv_stats_scratch_doubles() doubles, zeroed ONCE by the caller (the kernel
 * leaves it ready for the next call).  One launch; sums in fp64 in a fixed order. */
int64_t rp_adv_stats_scratch_doubles(void);
int rp_adv_stats_dev(const float *d_adv, const int64_t *d_index, int64_t batch, float *d_stats2, double *d_scratch, void *stream);
/* clip_grad_norm_(max_grad_norm) followed by Adam.step() as one launch: d_params, d_m, d_v are rp_grad_floats()
 * floats in the layout of the gradient vector d_grad (a policy whose parameters are views of d_params sees the
 * update in place); the gradient is first scaled by grad_scale (1 / world size after an all-reduce) and the entropy
 * bonus of the state-independent log-std (-ent_coef per component) is added; the loss slots are left alone.  step =
 * the number of this update, from 1 (bias corrections); eps enters as in torch.optim.Adam. */
int rp_clip_adam_dev(float *d_params, const float *d_grad, float *d_m, float *d_v, int obs_dim, int act_dim, float lr,
                     float beta1, float beta2, float eps, int64_t step, float max_grad_norm, float grad_scale, float ent_coef,
                     void *stream);
/* test hook: would a grant of lds_bytes of dynamic LDS be issued for (kernel id, device) now?  Records it. */
int rp_debug_lds_grant_needed(int kernel_id, int dev, int64_t lds_bytes);
/* which form of the gradient kernels rp_ppo_grad_dev launches for this policy: 2 = the small instance (obs_dim <= 31, up to 8
 * actions) with its inputs prefetched by LDS-DMA, 1 = the small instance loading at the start of each tile (the two input
 * buffers per wave do not fit beside the operands, or ROBOY_POLICY_PREFETCH=0), 0 = the general instance; < 0: unsupported */
int rp_grad_form(int obs_dim, int act_dim);

#ifdef __cplusplus
}
#endif
#endif
