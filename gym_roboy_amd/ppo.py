"""PPO (clipped surrogate) with an MLP policy, on torch-ROCm.

Consumer of the env: the reference trains with
stable_baselines' TF1 ``PPO2("MlpPolicy", env, ent_coef=0.1)`` over a
``SubprocVecEnv`` (``/root/reference/gym_roboy/train_parallel.py:28-35``).  Here
the rollout never leaves the GPU: ``RoboyVecEnv.step`` takes the policy's action
tensor and returns observation / reward / done tensors on the same HIP stream.
Hyper-parameters default to stable_baselines PPO2's (n_steps 128, 4 minibatches,
4 epochs, gamma 0.99, lambda 0.95, lr 2.5e-4, clip 0.2, vf 0.5, max grad norm
0.5) with the reference's ``ent_coef = 0.1``.  With several ranks (one per GPU)
gradients are averaged with ``torch.distributed.all_reduce`` (RCCL over xGMI).

``use_graphs=True`` (device env) captures a whole rollout - policy forward, sampling, the fused
env kernel launched through the C ABI, GAE - in one HIP graph and replays it, on every rank of a
multi-rank run (the rollout is rank-local), instead of launching ~30 small kernels per env step
from Python: at 4 096 envs 375 -> 102 us per vectorised step (``tools/ppo_profile.py``).

On a GPU the policy's work runs in the kernels of ``include/roboy_policy.h`` by default (exact f32
on the matrix cores): the policy step and GAE of the rollout (``FusedPolicyStep``, ``gae_fused``),
the minibatch gradient (``FusedPolicyGrad``) and the rest of the update - the epoch's sample order
as a keyed bijection evaluated on the device, the minibatch's advantage statistics applied inside
the gradient kernel, ``clip_grad_norm_`` + ``Adam.step`` as one launch over a flat parameter buffer
the module's parameters are views of (``FusedAdam``); with several ranks that flat gradient vector
is all-reduced once per minibatch.  At 262 144 envs a PPO iteration goes from 49 ms + 1.06 s
(rollout + update, torch: ~40 memory-bound passes over [8.4 M x 64] activations per minibatch) to
11 ms + 0.11 s.  ``fused_policy=False, fused_update=False`` select the torch path, which is what
the kernels are tested against.
"""
import math

import torch
from torch import nn


class MlpPolicy(nn.Module):
    """Two tanh layers of 64 units for the policy and for the value function,
    diagonal Gaussian with a state-independent log-std (stable_baselines' MlpPolicy)."""

    def __init__(self, obs_dim: int, act_dim: int, hidden: int = 64):
        super().__init__()
        def mlp(out):
            return nn.Sequential(nn.Linear(obs_dim, hidden), nn.Tanh(), nn.Linear(hidden, hidden), nn.Tanh(),
                                 nn.Linear(hidden, out))
        self.pi, self.vf = mlp(act_dim), mlp(1)
        self.log_std = nn.Parameter(torch.zeros(act_dim))
        for m in self.modules():
            if isinstance(m, nn.Linear):
                nn.init.orthogonal_(m.weight, math.sqrt(2))
                nn.init.zeros_(m.bias)
        nn.init.orthogonal_(self.pi[-1].weight, 0.01)
        nn.init.orthogonal_(self.vf[-1].weight, 1.0)

    def dist(self, obs):
        # validate_args=False: the check reads a device flag back (a host sync per call)
        return torch.distributions.Normal(self.pi(obs), self.log_std.exp(), validate_args=False)

    def value(self, obs):
        return self.vf(obs).squeeze(-1)

    @torch.no_grad()
    def act(self, obs, deterministic=False):
        d = self.dist(obs)
        # mean + std * eps rather than d.sample(): torch.normal(mean, std) checks std >= 0 on the host
        a = d.mean if deterministic else d.mean + d.stddev * torch.randn_like(d.mean)
        return a, d.log_prob(a).sum(-1), self.value(obs)


def gae(rewards, values, dones, last_value, gamma, lam):
    """Generalised advantage estimation over a [T, N] rollout (dones[t] = the
    step t ended an episode)."""
    T = rewards.shape[0]
    adv = torch.zeros_like(rewards)
    last = torch.zeros_like(last_value)
    next_value = last_value
    for t in range(T - 1, -1, -1):
        nonterminal = 1.0 - dones[t]
        delta = rewards[t] + gamma * next_value * nonterminal - values[t]
        last = delta + gamma * lam * nonterminal * last
        adv[t] = last
        next_value = values[t]
    return adv, adv + values


def gae_fused(rewards, values, dones, last_value, gamma, lam, adv_out=None, ret_out=None):
    """gae() as one kernel (include/roboy_policy.h: rp_gae_dev): the [T, N] tensors must be contiguous fp32 on the GPU."""
    import ctypes as c
    from . import _policy_native as pn
    T, N = rewards.shape
    adv = torch.empty_like(rewards) if adv_out is None else adv_out
    ret = torch.empty_like(rewards) if ret_out is None else ret_out
    ptr = lambda t: c.c_void_p(t.data_ptr())
    pn.check(pn.load().rp_gae_dev(ptr(rewards), ptr(values), ptr(dones), ptr(last_value), float(gamma), float(lam), ptr(adv), ptr(ret),
                                  int(T), int(N), c.c_void_p(torch.cuda.current_stream(rewards.device).cuda_stream)))
    return adv, ret


class FusedPolicyStep:
    """``MlpPolicy.act`` as one kernel on the matrix cores (include/roboy_policy.h, csrc/mlp_policy.hip): observation
    -> action sample, log-probability, value, written straight into the rollout buffers.  The parameters stay torch
    tensors (the optimiser - torch's or ``FusedAdam`` - updates them in place); ``pack()`` is one device-side gather into the operand order the
    kernel reads (the gather map depends on the dimensions only and is built once on the host).  Exploration noise is
    the library's Philox stream keyed (seed; sample, step), not torch's generator."""

    def __init__(self, policy, seed=0):
        import ctypes
        from . import _policy_native as pn
        self._pn, self._ct, self._lib = pn, ctypes, pn.load()
        self.policy = policy
        self.obs_dim, self.act_dim = policy.pi[0].in_features, policy.pi[-1].out_features
        if policy.pi[0].out_features != 64 or len(policy.pi) != 5:
            raise ValueError("the fused policy step is built for MlpPolicy's two hidden layers of 64 units")
        dev = policy.log_std.device
        m, total = pn.gather_map(self.obs_dim, self.act_dim)
        self._map = torch.from_numpy(m).to(dev)
        self._zero = torch.zeros(1, device=dev)
        self.seed = int(seed)

    def _params(self):
        pi, vf = self.policy.pi, self.policy.vf
        return [pi[0].weight, pi[0].bias, pi[2].weight, pi[2].bias, pi[4].weight, pi[4].bias,
                vf[0].weight, vf[0].bias, vf[2].weight, vf[2].bias, vf[4].weight, vf[4].bias, self.policy.log_std]

    @torch.no_grad()
    def pack(self):
        flat = torch.cat([p.detach().reshape(-1) for p in self._params()] + [self._zero])
        return flat[self._map]

    @torch.no_grad()
    def act_into(self, obs, act, logp, val, step=0, packed=None, mean=None, sample_offset=0, deterministic=False,
                 step_base=None):
        """obs [n, obs_dim] (contiguous, fp32, cuda) -> act [n, act_dim], logp [n], val [n] (preallocated)."""
        c = self._ct
        packed = self.pack() if packed is None else packed
        ptr = lambda t: c.c_void_p(t.data_ptr()) if t is not None else None
        self._pn.check(self._lib.rp_act_dev(
            ptr(packed), ptr(obs), ptr(act), ptr(logp), ptr(val), ptr(mean), int(obs.shape[0]), self.obs_dim, self.act_dim,
            self.seed, int(sample_offset), int(step), ptr(step_base), int(bool(deterministic)),
            c.c_void_p(torch.cuda.current_stream(obs.device).cuda_stream)))
        return packed


class FusedPolicyGrad:
    """The PPO minibatch gradient of an ``MlpPolicy`` as MFMA kernels (include/roboy_policy.h: rp_ppo_grad_dev,
    csrc/mlp_train.hip): forward, clipped-surrogate / clipped-value loss derivative, back-propagation and the weight
    gradients of both networks without materialising an activation in memory.  ``run()`` fills ``p.grad`` of every
    parameter (views of one flat buffer) and returns the two loss terms.  The advantage's per-minibatch normalisation
    happens inside (``adv_stats`` from ``minibatch_adv_stats()``); clipping and the optimiser step are ``FusedAdam``."""

    def __init__(self, policy):
        import ctypes
        from . import _policy_native as pn
        self._pn, self._ct, self._lib = pn, ctypes, pn.load()
        self.policy = policy
        self.obs_dim, self.act_dim = policy.pi[0].in_features, policy.pi[-1].out_features
        # the gather map and the gradient layout are built for MlpPolicy's shape: anything else would be indexed wrongly
        if (len(policy.pi) != 5 or len(policy.vf) != 5 or policy.pi[0].out_features != 64 or policy.pi[2].out_features != 64
                or policy.vf[0].out_features != 64 or policy.vf[2].out_features != 64):
            raise ValueError("the fused PPO gradient is built for MlpPolicy's two hidden layers of 64 units")
        dev = policy.log_std.device
        m, _ = pn.gather_map(self.obs_dim, self.act_dim, train=True)
        self._map = torch.from_numpy(m).to(dev)
        self._zero = torch.zeros(1, device=dev)
        layout, n = pn.grad_layout(self.obs_dim, self.act_dim)
        self._g = torch.zeros(n, device=dev)
        self._layout = layout
        pi, vf = policy.pi, policy.vf
        self._named = {"pi_w1": pi[0].weight, "pi_b1": pi[0].bias, "pi_w2": pi[2].weight, "pi_b2": pi[2].bias,
                       "pi_w3": pi[4].weight, "pi_b3": pi[4].bias, "vf_w1": vf[0].weight, "vf_b1": vf[0].bias,
                       "vf_w2": vf[2].weight, "vf_b2": vf[2].bias, "vf_w3": vf[4].weight, "vf_b3": vf[4].bias,
                       "log_std": policy.log_std}
        self._views = {}
        for name, p in self._named.items():
            off, shape = layout[name]
            self._views[name] = self._g[off:off + p.numel()].view(shape)
        self._ws = None
        self._stats = torch.zeros(2, device=dev)
        self._stat_scratch = torch.zeros(int(self._lib.rp_adv_stats_scratch_doubles()), dtype=torch.float64, device=dev)

    @torch.no_grad()
    def minibatch_adv_stats(self, adv_full, index):
        """{mean, 1 / (std + 1e-8)} of adv_full[index] (device tensor of 2 floats, overwritten by the next call)."""
        c = self._ct
        self._pn.check(self._lib.rp_adv_stats_dev(
            c.c_void_p(adv_full.data_ptr()), c.c_void_p(index.data_ptr()) if index is not None else None,
            int(index.shape[0] if index is not None else adv_full.shape[0]), c.c_void_p(self._stats.data_ptr()),
            c.c_void_p(self._stat_scratch.data_ptr()), c.c_void_p(torch.cuda.current_stream(adv_full.device).cuda_stream)))
        return self._stats

    @torch.no_grad()
    def run(self, obs, act, adv, logp_old, val_old, ret, cliprange, vf_coef, ent_coef, index=None, adv_stats=None,
            entropy_grad=True):
        """index (int64 [B], optional): minibatch sample i is row index[i] of obs / act / logp_old / val_old / ret
        (the whole rollout's tensors, no gathered copies).  adv: in minibatch order and normalised by the caller, or
        - with adv_stats (minibatch_adv_stats()) - the rollout's raw advantage, indexed like the rest and normalised
        in the kernel.  entropy_grad=False leaves the entropy bonus of the log-std to FusedAdam.step()."""
        c = self._ct
        B = int(index.shape[0]) if (index is not None and adv_stats is not None) else int(adv.shape[0])
        flat = torch.cat([self._named[k].detach().reshape(-1) for k in self._pn.PARAM_ORDER] + [self._zero])
        packed = flat[self._map]
        need = int(self._lib.rp_ppo_workspace_floats(self.obs_dim, self.act_dim, B))
        if self._ws is None or self._ws.numel() < need:
            self._ws = torch.empty(need, device=obs.device)
        ptr = lambda t: c.c_void_p(t.data_ptr()) if t is not None else None
        self._pn.check(self._lib.rp_ppo_grad_dev(
            ptr(packed), ptr(obs), ptr(act), ptr(adv), ptr(adv_stats), ptr(logp_old), ptr(val_old), ptr(ret), ptr(index), B,
            self.obs_dim, self.act_dim,
            float(cliprange), float(vf_coef), ptr(self._g), ptr(self._ws),
            c.c_void_p(torch.cuda.current_stream(obs.device).cuda_stream)))
        if entropy_grad:
            self._views["log_std"] -= ent_coef        # entropy bonus of a state-independent log-std
        for name, p in self._named.items():
            p.grad = self._views[name]
        pg = self._g[self._layout["pi_loss"][0]]
        vf = self._g[self._layout["vf_loss"][0]]
        return pg, vf


class FusedAdam:
    """``clip_grad_norm_`` + ``torch.optim.Adam.step`` as ONE launch (include/roboy_policy.h: rp_clip_adam_dev) over the
    flat gradient vector of a ``FusedPolicyGrad``.  The policy's parameters become views of one flat buffer laid out
    like that vector, so the kernel updates them in place and the cross-rank average is one all-reduce of one
    contiguous tensor.  State (first / second moments, step count) is checkpointed by ``state_dict()``."""

    def __init__(self, fgrad, lr, betas=(0.9, 0.999), eps=1e-5, max_grad_norm=0.5):
        self._f = fgrad
        self.lr, self.betas, self.eps, self.max_grad_norm = float(lr), (float(betas[0]), float(betas[1])), float(eps), float(max_grad_norm)
        g = fgrad._g
        self.params = torch.zeros_like(g)
        with torch.no_grad():
            for name, p in fgrad._named.items():
                off, shape = fgrad._layout[name]
                view = self.params[off:off + p.numel()].view(shape)
                view.copy_(p.data)
                p.data = view                              # the module's parameter IS the slice of the flat buffer
        self.m, self.v = torch.zeros_like(g), torch.zeros_like(g)
        self.t = 0

    @torch.no_grad()
    def step(self, ent_coef, dist=None):
        f, c = self._f, self._f._ct
        scale = 1.0
        if dist is not None and dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            if dist.get_backend() == "nccl":
                dist.all_reduce(f._g)                      # one bucket: the whole gradient vector, contiguous
            else:                                          # rehearsal over gloo: through the host
                host = f._g.cpu()
                dist.all_reduce(host)
                f._g.copy_(host)
            scale = 1.0 / dist.get_world_size()
        self.t += 1
        ptr = lambda t: c.c_void_p(t.data_ptr())
        f._pn.check(f._lib.rp_clip_adam_dev(ptr(self.params), ptr(f._g), ptr(self.m), ptr(self.v), f.obs_dim, f.act_dim, self.lr,
                                            self.betas[0], self.betas[1], self.eps, self.t, self.max_grad_norm, scale, float(ent_coef),
                                            c.c_void_p(torch.cuda.current_stream(f._g.device).cuda_stream)))
        return scale

    def state_dict(self):
        return {"fused_adam": True, "m": self.m.clone(), "v": self.v.clone(), "t": self.t}

    def load_state_dict(self, sd):
        if tuple(sd["m"].shape) != tuple(self.m.shape) or tuple(sd["v"].shape) != tuple(self.v.shape):
            raise ValueError("optimiser state of another policy layout: moments of %d values, this policy's flat vector has %d "
                             "(obs_dim %d, act_dim %d)" % (sd["m"].numel(), self.m.numel(), self._f.obs_dim, self._f.act_dim))
        self.m.copy_(sd["m"]); self.v.copy_(sd["v"]); self.t = int(sd["t"])

    def rebind(self, state_dict):
        """load_state_dict() of the module copies INTO the views (in place), so nothing to re-point; kept for clarity."""
        return state_dict


def _flat_names(policy):
    """layout name (include/roboy_policy.h: the flat gradient / parameter vector) -> parameter of an MlpPolicy"""
    pi, vf = policy.pi, policy.vf
    return {"pi_w1": pi[0].weight, "pi_b1": pi[0].bias, "pi_w2": pi[2].weight, "pi_b2": pi[2].bias, "pi_w3": pi[4].weight,
            "pi_b3": pi[4].bias, "vf_w1": vf[0].weight, "vf_b1": vf[0].bias, "vf_w2": vf[2].weight, "vf_b2": vf[2].bias,
            "vf_w3": vf[4].weight, "vf_b3": vf[4].bias, "log_std": policy.log_std}


def _param_slices(policy, layout):
    """[(offset, shape)] in the flat vector for policy.parameters(), in that order"""
    by_id = {id(p): name for name, p in _flat_names(policy).items()}
    return [layout[by_id[id(p)]] for p in policy.parameters()]


def adam_state_torch_to_flat(sd, policy, layout, m, v):
    """Moments and step count of a ``torch.optim.Adam`` state_dict over ``policy.parameters()`` into the flat vectors m, v of
    the fused optimiser; returns the step count, or None if the state holds no moments yet (an optimiser that never stepped)."""
    state = sd.get("state", {})
    if not state:
        return None
    order = [i for g in sd["param_groups"] for i in g["params"]]
    slices = _param_slices(policy, layout)
    if len(order) != len(slices):
        raise ValueError("optimiser state of another module: %d parameters, this policy has %d" % (len(order), len(slices)))
    # every parameter or none: moments of some parameters under ONE step count would resume the others with zero moments and a
    # bias correction that assumes they have been stepping all along
    # (a parameter with requires_grad = False is never stepped by torch.optim.Adam and has no state: zero moments are exact for it)
    frozen = {idx for idx, p in zip(order, policy.parameters()) if not p.requires_grad}
    missing = [idx for idx in order if state.get(idx) is None and idx not in frozen]
    if missing:
        raise ValueError("partial optimiser state: %d of %d parameters carry no moments (a torch.optim.Adam that stepped only some of "
                         "its parameters cannot be converted to the flat form)" % (len(missing), len(order)))
    t = 0
    for idx, (off, shape) in zip(order, slices):
        st = state.get(idx)
        if st is None:                                       # frozen: leave its (zero) moments alone
            continue
        if tuple(st["exp_avg"].shape) != tuple(shape):
            raise ValueError("optimiser state of another policy layout: %r against %r" % (tuple(st["exp_avg"].shape), tuple(shape)))
        n = int(st["exp_avg"].numel())
        m[off:off + n].copy_(st["exp_avg"].reshape(-1)); v[off:off + n].copy_(st["exp_avg_sq"].reshape(-1))
        t = max(t, int(st["step"]))
    return t


def adam_state_flat_to_torch(fsd, policy, layout, template_sd):
    """A ``FusedAdam`` state (flat moments m, v and step count t) as the state_dict of a ``torch.optim.Adam`` over
    ``policy.parameters()`` with the parameter groups of ``template_sd``."""
    order = [i for g in template_sd["param_groups"] for i in g["params"]]
    state = {}
    for idx, (off, shape) in zip(order, _param_slices(policy, layout)):
        n = 1
        for d in shape:
            n *= int(d)
        state[idx] = {"step": torch.tensor(float(fsd["t"])), "exp_avg": fsd["m"][off:off + n].view(shape).clone(),
                      "exp_avg_sq": fsd["v"][off:off + n].view(shape).clone()}
    return {"state": state, "param_groups": template_sd["param_groups"]}


def average_gradients(module, dist=None):
    if dist is None or not dist.is_available() or not dist.is_initialized() or dist.get_world_size() == 1:
        return
    world = dist.get_world_size()
    flat = torch.cat([p.grad.reshape(-1) for p in module.parameters() if p.grad is not None])
    dist.all_reduce(flat)          # one bucket: the policy is ~10^4 parameters
    flat /= world
    off = 0
    for p in module.parameters():
        if p.grad is not None:
            n = p.grad.numel()
            p.grad.copy_(flat[off:off + n].view_as(p.grad))
            off += n


def _fused_kernels_apply(policy, obs_dim, act_dim):
    """MlpPolicy's shape (two hidden layers of 64 units per net) in dimensions the kernels of include/roboy_policy.h
    support, and the library is there."""
    try:
        from . import _policy_native as pn
        ok_shape = (isinstance(policy, MlpPolicy) and len(policy.pi) == 5 and len(policy.vf) == 5
                    and all(l.out_features == 64 for l in (policy.pi[0], policy.pi[2], policy.vf[0], policy.vf[2])))
        return bool(ok_shape and pn.load().rp_train_packed_floats(obs_dim, act_dim) > 0)
    except Exception:
        return False


class PPO:
    def __init__(self, env, policy=None, n_steps=128, nminibatches=4, noptepochs=4, gamma=0.99, lam=0.95,
                 learning_rate=2.5e-4, cliprange=0.2, ent_coef=0.01, vf_coef=0.5, max_grad_norm=0.5,
                 device="cuda", dist=None, reward_scale=1.0, seed=0, use_graphs=False, fused_policy=None,
                 fused_update=None, rollout_chains=None):
        """fused_policy / fused_update: None = the fused MFMA kernels whenever they apply (a GPU, MlpPolicy's shape,
        dimensions the kernels support), True = insist, False = the torch path (the statement the kernels are
        tested against).
        rollout_chains: graph mode with the fused policy step - 2 = the rollout as two independent chains of (policy step, env
        step) launches over the two halves of the batch on two streams, joined in front of GAE (the sub-range entry points of
        include/roboy_sim.h: one half's launch gaps and load / store phases lie under the other half's kernels; the results do
        not depend on the split); 1 = one chain over the whole batch; None = two from ``CHAIN_BATCH`` envs on where the env's
        kernel form steps sub-ranges and the policy is of MsjRobot's size."""
        self.env, self.dist, self.device = env, dist, torch.device(device)
        self._chains_arg = rollout_chains
        torch.manual_seed(seed)
        obs_dim = env.observation_space.shape[0]
        act_dim = env.action_space.shape[0]
        self.policy = (policy or MlpPolicy(obs_dim, act_dim)).to(self.device)
        if dist is not None and dist.is_available() and dist.is_initialized():
            for p in self.policy.parameters():          # same initial weights on every rank
                dist.broadcast(p.data, 0)
            # ... but each rank's own exploration noise and minibatch order: the reference seeds worker
            # `rank` with seed + rank (/root/reference/gym_roboy/train_parallel.py:24)
            torch.manual_seed(seed + dist.get_rank())
        multi_rank = dist is not None and dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1
        # the rollout (policy step, env step, GAE) is rank-local, so it is captured on every rank; the collectives
        # (gradient average, statistics) stay outside the graph
        self.use_graphs = bool(use_graphs) and self.device.type == "cuda" and hasattr(env, "step_dev")
        if fused_policy is None or fused_update is None:
            auto = self.device.type == "cuda" and _fused_kernels_apply(self.policy, obs_dim, act_dim)
            fused_policy = auto if fused_policy is None else fused_policy
            fused_update = auto if fused_update is None else fused_update
        self.opt = torch.optim.Adam(self.policy.parameters(), lr=learning_rate, eps=1e-5)
        self._epoch = 0                       # update epochs so far: keys the fused path's sample order
        self._seed = int(seed) + (7919 * dist.get_rank() if multi_rank else 0)
        self._rollout_graph = None
        self.rollout_chains = 1
        # fused_policy: the rollout's policy step runs as one MFMA kernel (FusedPolicyStep) instead of ~30 torch kernels
        self._fused = None
        if fused_policy:
            if self.device.type != "cuda":
                raise ValueError("fused_policy needs a GPU")
            rank = dist.get_rank() if multi_rank else 0
            self._fused = FusedPolicyStep(self.policy, seed=seed + 7919 * rank)
            self._step_base = torch.zeros(1, dtype=torch.int32, device=self.device)    # rollout steps taken so far
        # fused_update: the minibatch gradient comes from the MFMA kernels (FusedPolicyGrad) instead of torch autograd
        self._fgrad = None
        if fused_update:
            if self.device.type != "cuda":
                raise ValueError("fused_update needs a GPU")
            self._fgrad = FusedPolicyGrad(self.policy)
            self._fadam = FusedAdam(self._fgrad, learning_rate, eps=1e-5, max_grad_norm=max_grad_norm)
        self.n_steps, self.nminibatches, self.noptepochs = n_steps, nminibatches, noptepochs
        self.gamma, self.lam, self.cliprange = gamma, lam, cliprange
        self.ent_coef, self.vf_coef, self.max_grad_norm = ent_coef, vf_coef, max_grad_norm
        self.reward_scale = reward_scale
        self.num_timesteps = 0
        self._obs = None

    def _to_tensor(self, x, dtype=torch.float32):
        return x.to(self.device, dtype) if torch.is_tensor(x) else torch.as_tensor(x, dtype=dtype, device=self.device)

    # -- HIP-graph mode ------------------------------------------------------------
    # two chains from this many envs on, for policies of MsjRobot's size (measured, us per vectorised rollout step, one -> two chains,
    # profiles/r4_a/ppo_chains.log: MsjRobot 32 768 envs 21.7 -> 25.8, 65 536 envs 31.5 -> 28.4, 262 144 envs 87.6 -> 76.9, 1 M envs
    # 349 -> 296; the upper body's 60 -> 38 policy step does not share the chip with a second launch: 65 536 envs 92.8 -> 91.5,
    # 32 768 envs 55.6 -> 101.8)
    CHAIN_BATCH = 65536
    CHAIN_MAX_OBS = 29

    def _rollout_steps(self, b, lo, hi, stream_ptr=None):
        """The T (policy step, env step) pairs of envs [lo, hi): the whole batch on the env's stream (stream_ptr None), or one
        chain's half on its own stream."""
        env, T = self.env, self.n_steps
        whole = stream_ptr is None
        b["obs"][0][lo:hi].copy_(b["carry"][lo:hi])
        packed = self._fused.pack() if self._fused is not None else None     # (inside the graph: re-gathered on every replay)
        for t in range(T):
            if self._fused is not None:
                # straight into the rollout buffers; the env kernel clamps the action to its box itself.  The noise is keyed by
                # the GLOBAL sample index (sample_offset): the same draw however the batch is cut
                self._fused.act_into(b["obs"][t][lo:hi], b["act"][t][lo:hi], b["logp"][t][lo:hi], b["val"][t][lo:hi], step=t,
                                     packed=packed, step_base=self._step_base, sample_offset=lo)
                if whole:
                    env.step_dev(b["act"][t].data_ptr(), b["obs"][t + 1].data_ptr(), b["rew_raw"][t].data_ptr(),
                                 b["done_i"][t].data_ptr())
                else:
                    env.step_range_dev(lo, hi - lo, stream_ptr, b["act"][t].data_ptr(), b["obs"][t + 1].data_ptr(),
                                       b["rew_raw"][t].data_ptr(), b["done_i"][t].data_ptr())
                continue
            a, logp, v = self.policy.act(b["obs"][t])
            b["act"][t].copy_(a); b["logp"][t].copy_(logp); b["val"][t].copy_(v)
            clipped = a.clamp(-1.0, 1.0).contiguous()
            env.step_dev(clipped.data_ptr(), b["obs"][t + 1].data_ptr(), b["rew_raw"][t].data_ptr(),
                         b["done_i"][t].data_ptr())

    def _rollout_tail(self, b):
        T = self.n_steps
        if self._fused is not None:
            self._step_base += T                                             # fresh noise on the next replay
        b["rew"].copy_(b["rew_raw"] * self.reward_scale)
        b["done"].copy_(b["done_i"].to(torch.float32))
        with torch.no_grad():
            last_value = self.policy.value(b["obs"][T])
        if self._fused is not None:
            gae_fused(b["rew"], b["val"], b["done"], last_value.contiguous(), self.gamma, self.lam, b["adv"], b["ret"])
        else:
            adv, ret = gae(b["rew"], b["val"], b["done"], last_value, self.gamma, self.lam)
            b["adv"].copy_(adv); b["ret"].copy_(ret)
        b["carry"].copy_(b["obs"][T])

    def _rollout_body(self, b):
        self._rollout_steps(b, 0, b["carry"].shape[0])
        self._rollout_tail(b)

    def _pick_chains(self, N):
        asked = self._chains_arg is not None and int(self._chains_arg) >= 2
        why = None
        if self._fused is None:
            why = "the fused policy step is off (fused_policy=False)"
        elif not hasattr(self.env, "step_range_dev") or not self.env.range_capable():
            why = "the env's kernel form steps whole batches only"
        elif asked and N < 512:
            why = "fewer than 512 envs"
        if why is not None:
            if asked:      # an explicit request that cannot be honoured is said, not dropped silently
                import warnings
                warnings.warn("rollout_chains=%r not honoured (%s): the rollout runs as one chain" % (self._chains_arg, why), RuntimeWarning, stacklevel=3)
            return 1
        if self._chains_arg is not None:
            return 2 if asked else 1
        return 2 if N >= self.CHAIN_BATCH and self._fused.obs_dim <= self.CHAIN_MAX_OBS else 1

    def _build_rollout_graph(self):
        env, T, dev = self.env, self.n_steps, self.device
        N, od, ad = env.num_envs, env.observation_space.shape[0], env.action_space.shape[0]
        z = lambda *shape, dtype=torch.float32: torch.zeros(shape, dtype=dtype, device=dev)
        b = {"obs": z(T + 1, N, od), "act": z(T, N, ad), "logp": z(T, N), "val": z(T, N), "rew_raw": z(T, N),
             "rew": z(T, N), "done_i": z(T, N, dtype=torch.int32), "done": z(T, N), "adv": z(T, N), "ret": z(T, N),
             "carry": z(N, od)}
        b["carry"].copy_(self._to_tensor(env.reset()) if self._obs is None else self._obs)
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        # the env kernel is launched on the simulation's stream: make that the capture stream
        env.set_stream(side.cuda_stream)
        # kernels the simulation specialises at run time (hiprtc) are built now: a compilation and a module load
        # must not fall into the capture below
        if hasattr(getattr(env, "sim", None), "specialization"):
            env.sim.specialization()
        with torch.cuda.stream(side), torch.no_grad():
            # first use of the GEMM library for these shapes (handle, workspace) must not fall into the capture
            self.policy.act(b["carry"]); self.policy.value(b["carry"])
            if self._fused is not None:       # its one-time launch configuration must not fall into the capture either
                self._fused.act_into(b["carry"], b["act"][0], b["logp"][0], b["val"][0], deterministic=True)
        side.synchronize()
        self.rollout_chains = self._pick_chains(N)
        # thread-local capture mode: another thread of the process (RCCL's watchdog in a multi-rank run) may call into
        # the runtime while this one captures
        if self.rollout_chains == 1:
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph, stream=side, capture_error_mode="thread_local"):
                self._rollout_body(b)
            self._rollout_graph = graph
        else:
            # two chains: one graph per half of the batch (each on a stream of its own when replayed) and one for what follows
            # the join.  Two LINEAR graphs, not one with two branches: csrc/roboy_sim.hip, rb_rollout_dev.
            mid = ((N // 2 + 255) // 256) * 256
            side2 = torch.cuda.Stream(device=dev)
            self._chain_stream = side2
            graphs = []
            for (lo, hi), st in (((0, mid), side), ((mid, N), side2)):
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, stream=st, capture_error_mode="thread_local"):
                    self._rollout_steps(b, lo, hi, stream_ptr=st.cuda_stream)
                graphs.append(g)
            tail = torch.cuda.CUDAGraph()
            with torch.cuda.graph(tail, stream=side, capture_error_mode="thread_local"):
                self._rollout_tail(b)
            self._rollout_graph = (graphs[0], graphs[1], tail)
        torch.cuda.current_stream(dev).wait_stream(side)
        # replays (and every later eager env call) run on the caller's current stream
        env.set_stream(torch.cuda.current_stream(dev).cuda_stream)
        if hasattr(env, "note_replayed_steps"):
            env.note_replayed_steps(-T)    # the capture pass went through the counting entry point without running
        self._rb = b

    def _collect_graph(self):
        if self._rollout_graph is None:
            self._build_rollout_graph()
        if self.rollout_chains == 1:
            self._rollout_graph.replay()
        else:
            # fork - the second chain on its own stream, launched first (rb_rollout_dev's order) - replay, join, tail
            ga, gb, tail = self._rollout_graph
            cur = torch.cuda.current_stream(self.device)
            self._chain_stream.wait_stream(cur)
            with torch.cuda.stream(self._chain_stream):
                gb.replay()
            ga.replay()
            cur.wait_stream(self._chain_stream)
            tail.replay()
        b, T = self._rb, self.n_steps
        if hasattr(self.env, "note_replayed_steps"):
            self.env.note_replayed_steps(T)
        self._obs = b["carry"]
        self.num_timesteps += T * b["carry"].shape[0]
        return {"obs": b["obs"][:T], "act": b["act"], "logp": b["logp"], "val": b["val"], "rew": b["rew"],
                "done": b["done"], "adv": b["adv"], "ret": b["ret"]}

    def collect(self):
        if self.use_graphs:
            return self._collect_graph()
        env, T = self.env, self.n_steps
        if self._obs is None:
            self._obs = self._to_tensor(env.reset())
        N = self._obs.shape[0]
        buf = {k: [] for k in ("obs", "act", "logp", "val", "rew", "done")}
        packed = self._fused.pack() if self._fused is not None else None
        for t in range(T):
            if self._fused is not None:
                o = self._obs.contiguous()
                a = torch.empty(N, self._fused.act_dim, device=self.device)
                logp, v = torch.empty(N, device=self.device), torch.empty(N, device=self.device)
                self._fused.act_into(o, a, logp, v, step=t, packed=packed, step_base=self._step_base)
            else:
                a, logp, v = self.policy.act(self._obs)
            clipped = a.clamp(-1.0, 1.0).contiguous()        # the env's action box (roboy_env.py:31)
            obs, rew, done, _ = env.step(clipped if self.device.type == "cuda" else clipped.cpu().numpy())
            buf["obs"].append(self._obs); buf["act"].append(a); buf["logp"].append(logp); buf["val"].append(v)
            buf["rew"].append(self._to_tensor(rew) * self.reward_scale)
            buf["done"].append(self._to_tensor(done))
            self._obs = self._to_tensor(obs)
        if self._fused is not None:
            self._step_base += T
        roll = {k: torch.stack(v) for k, v in buf.items()}
        with torch.no_grad():
            last_value = self.policy.value(self._obs)
        roll["adv"], roll["ret"] = gae(roll["rew"], roll["val"], roll["done"], last_value, self.gamma, self.lam)
        self.num_timesteps += T * N
        return roll

    def _minibatch_loss(self, flat, idx):
        obs, act = flat["obs"][idx], flat["act"][idx]
        adv = flat["adv"][idx]
        adv = (adv - adv.mean()) / (adv.std() + 1e-8)
        d = self.policy.dist(obs)
        logp = d.log_prob(act).sum(-1)
        ratio = (logp - flat["logp"][idx]).exp()
        pg = torch.max(-adv * ratio, -adv * ratio.clamp(1 - self.cliprange, 1 + self.cliprange)).mean()
        v = self.policy.value(obs)
        v_clip = flat["val"][idx] + (v - flat["val"][idx]).clamp(-self.cliprange, self.cliprange)
        vf = 0.5 * torch.max((v - flat["ret"][idx]) ** 2, (v_clip - flat["ret"][idx]) ** 2).mean()
        ent = d.entropy().sum(-1).mean()
        return pg - self.ent_coef * ent + self.vf_coef * vf, pg, vf, ent

    def _minibatch_step_fused(self, flat, idx):
        """Four launches + two small reductions per minibatch: advantage statistics, the two gradient kernels (which
        gather the rollout's rows through idx and normalise the advantage per sample), clip + Adam."""
        stats = self._fgrad.minibatch_adv_stats(flat["adv"], idx)
        pg, vf = self._fgrad.run(flat["obs"], flat["act"], flat["adv"], flat["logp"], flat["val"], flat["ret"], self.cliprange,
                                 self.vf_coef, self.ent_coef, index=idx, adv_stats=stats, entropy_grad=False)
        scale = self._fadam.step(self.ent_coef, self.dist)          # all-reduces the gradient vector first when ranks > 1
        ent = (0.5 + 0.5 * math.log(2 * math.pi) + self.policy.log_std.detach()).sum()
        pg, vf = pg * scale, vf * scale                              # (the loss slots were summed over the ranks with the rest)
        return pg - self.ent_coef * ent + self.vf_coef * vf, pg, vf, ent

    def _minibatch_step(self, flat, idx):
        if self._fgrad is not None:
            return self._minibatch_step_fused(flat, idx)
        loss, pg, vf, ent = self._minibatch_loss(flat, idx)
        self.opt.zero_grad(set_to_none=True)
        loss.backward()
        average_gradients(self.policy, self.dist)
        nn.utils.clip_grad_norm_(self.policy.parameters(), self.max_grad_norm)
        self.opt.step()
        return loss, pg, vf, ent

    def _sample_order(self, n):
        """The epoch's sample order.  Torch path: torch.randperm (a sort of n random keys).  Fused path: a keyed bijection
        of [0, n) evaluated per element on the device (rp_perm_dev), key = (seed, epochs so far)."""
        self._epoch += 1
        if self._fgrad is None:
            return torch.randperm(n, device=self.device)
        import ctypes as c
        if getattr(self, "_perm_buf", None) is None or self._perm_buf.numel() != n:
            self._perm_buf = torch.empty(n, dtype=torch.int64, device=self.device)
        key = ((self._seed & 0xFFFFFFFF) << 32) | (self._epoch & 0xFFFFFFFF)
        f = self._fgrad
        f._pn.check(f._lib.rp_perm_dev(key, n, 0, n, c.c_void_p(self._perm_buf.data_ptr()),
                                       c.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)))
        return self._perm_buf

    def update(self, roll, sample_orders=None):
        """sample_orders (optional): one int64 index tensor per epoch instead of the generated order (tests feed the
        same orders to both optimiser paths)."""
        flat = {k: v.reshape(-1, *v.shape[2:]) for k, v in roll.items()}
        if self._fgrad is not None:          # the gradient kernels index the rollout tensors directly
            flat = {k: v.contiguous() for k, v in flat.items()}
        n = flat["obs"].shape[0]
        mb = max(n // self.nminibatches, 1)
        out = None
        for ep in range(self.noptepochs):
            perm = self._sample_order(n) if sample_orders is None else sample_orders[ep].contiguous()
            for s in range(0, n - mb + 1, mb):
                out = self._minibatch_step(flat, perm[s:s + mb])
        if out is None:
            return {}
        loss, pg, vf, ent = out
        return {"loss": loss.item(), "pg_loss": pg.item(), "vf_loss": vf.item(), "entropy": ent.item()}

    def learn(self, total_timesteps, log=None):
        target = self.num_timesteps + total_timesteps
        while self.num_timesteps < target:
            roll = self.collect()
            stats = self.update(roll)
            stats["mean_reward"] = roll["rew"].mean().item() / self.reward_scale
            stats["timesteps"] = self.num_timesteps
            if log:
                log(stats)
        return self

    def save(self, path):
        opt = self._fadam.state_dict() if self._fgrad is not None else self.opt.state_dict()
        torch.save({"policy": self.policy.state_dict(), "optimizer": opt, "num_timesteps": self.num_timesteps,
                    "epoch": self._epoch}, path)

    def load(self, path):
        ck = torch.load(path, map_location=self.device)
        self.policy.load_state_dict(ck["policy"])      # copies in place: the views of the fused optimiser's flat buffer stay valid
        fused_ck = isinstance(ck["optimizer"], dict) and ck["optimizer"].get("fused_adam", False)
        if self._fgrad is not None and fused_ck:
            self._fadam.load_state_dict(ck["optimizer"])
        elif self._fgrad is None and not fused_ck:
            self.opt.load_state_dict(ck["optimizer"])
        elif self._fgrad is not None:
            # a checkpoint of the torch path (fused_update=False, or written before the fused update was the default): its
            # moments and step count move into the flat vectors
            t = adam_state_torch_to_flat(ck["optimizer"], self.policy, self._fgrad._layout, self._fadam.m, self._fadam.v)
            if t is None:
                import warnings
                warnings.warn("the checkpoint's optimiser had not stepped yet: Adam's moments start afresh")
            else:
                self._fadam.t = t
        else:
            # a checkpoint of the fused path resumed on the torch path: torch's per-parameter state from the flat moments
            from . import _policy_native as pn
            layout, n = pn.grad_layout(self.policy.pi[0].in_features, self.policy.pi[-1].out_features)
            if int(ck["optimizer"]["m"].numel()) != int(n):
                raise ValueError("optimiser state of another policy layout: %d values, this policy's flat vector has %d"
                                 % (ck["optimizer"]["m"].numel(), n))
            self.opt.load_state_dict(adam_state_flat_to_torch(ck["optimizer"], self.policy, layout, self.opt.state_dict()))
        self.num_timesteps = ck["num_timesteps"]
        self._epoch = int(ck.get("epoch", 0))
        if self._fused is not None:
            # the exploration noise is keyed (seed; sample, step): continue the step count where the run stopped, so that a
            # resumed run does not replay the noise of its first rollouts
            n_envs = getattr(self.env, "num_envs", 1)
            self._step_base.fill_(int(self.num_timesteps // max(n_envs, 1)) & 0x7FFFFFFF)
        return self
