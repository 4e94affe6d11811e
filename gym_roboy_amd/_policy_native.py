"""ctypes binding of include/roboy_policy.h (gym_roboy_amd/csrc/libroboy_policy.so): the PPO consumer's fused policy
step.  Like _native.py it fails loudly when the library is missing - there is no fallback path behind it (the torch
MlpPolicy is a different, explicitly selected code path of ppo.py, not a substitute the caller gets silently)."""
import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None
RP_ABI_VERSION = 2


class MlpParams(ctypes.Structure):
    _fields_ = [(n, ctypes.c_void_p) for n in (
        "pi_w1", "pi_b1", "pi_w2", "pi_b2", "pi_w3", "pi_b3", "vf_w1", "vf_b1", "vf_w2", "vf_b2", "vf_w3", "vf_b3", "log_std")]


SIGNATURES = {
    "rp_abi_version": (ctypes.c_int, []),
    "rp_last_error": (ctypes.c_char_p, []),
    "rp_packed_floats": (ctypes.c_int64, [ctypes.c_int, ctypes.c_int]),
    "rp_pack": (ctypes.c_int, [ctypes.POINTER(MlpParams), ctypes.c_int, ctypes.c_int, ctypes.c_void_p]),
    "rp_gae_dev": (ctypes.c_int, [ctypes.c_void_p] * 4 + [ctypes.c_float, ctypes.c_float, ctypes.c_void_p, ctypes.c_void_p,
                                  ctypes.c_int, ctypes.c_int64, ctypes.c_void_p]),
    "rp_train_packed_floats": (ctypes.c_int64, [ctypes.c_int, ctypes.c_int]),
    "rp_pack_train": (ctypes.c_int, [ctypes.POINTER(MlpParams), ctypes.c_int, ctypes.c_int, ctypes.c_void_p]),
    "rp_grad_floats": (ctypes.c_int64, [ctypes.c_int, ctypes.c_int]),
    "rp_ppo_workspace_floats": (ctypes.c_int64, [ctypes.c_int, ctypes.c_int, ctypes.c_int64]),
    "rp_ppo_grad_dev": (ctypes.c_int, [ctypes.c_void_p] * 9 + [ctypes.c_int64, ctypes.c_int, ctypes.c_int, ctypes.c_float,
                                       ctypes.c_float, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]),
    "rp_perm_dev": (ctypes.c_int, [ctypes.c_uint64, ctypes.c_int64, ctypes.c_int64, ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p]),
    "rp_perm_host": (ctypes.c_int, [ctypes.c_uint64, ctypes.c_int64, ctypes.c_int64, ctypes.c_int64, ctypes.c_void_p]),
    "rp_adv_stats_scratch_doubles": (ctypes.c_int64, []),
    "rp_adv_stats_dev": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]),
    "rp_clip_adam_dev": (ctypes.c_int, [ctypes.c_void_p] * 4 + [ctypes.c_int, ctypes.c_int] + [ctypes.c_float] * 4 +
                                        [ctypes.c_int64, ctypes.c_float, ctypes.c_float, ctypes.c_float, ctypes.c_void_p]),
    "rp_debug_lds_grant_needed": (ctypes.c_int, [ctypes.c_int, ctypes.c_int, ctypes.c_int64]),
    "rp_grad_form": (ctypes.c_int, [ctypes.c_int, ctypes.c_int]),
    "rp_act_dev": (ctypes.c_int, [ctypes.c_void_p] * 6 + [ctypes.c_int64, ctypes.c_int, ctypes.c_int, ctypes.c_uint64,
                                  ctypes.c_uint64, ctypes.c_uint32, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]),
}


def library_path():
    return os.environ.get("ROBOY_POLICY_LIB") or os.path.join(_HERE, "csrc", "libroboy_policy.so")


def load():
    global _LIB
    if _LIB is None:
        path = library_path()
        if not os.path.exists(path):
            raise RuntimeError("%s not found: build it with `make -C gym_roboy_amd/csrc` (or __graft_entry__.build())" % path)
        lib = ctypes.CDLL(path)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)
            fn.restype, fn.argtypes = res, args
        if lib.rp_abi_version() != RP_ABI_VERSION:
            raise RuntimeError("libroboy_policy.so has ABI version %d, this binding %d" % (lib.rp_abi_version(), RP_ABI_VERSION))
        _LIB = lib
    return _LIB


def check(rc):
    if rc != 0:
        raise RuntimeError("roboy_policy: %s (code %d)" % (load().rp_last_error().decode(), rc))


PARAM_ORDER = ("pi_w1", "pi_b1", "pi_w2", "pi_b2", "pi_w3", "pi_b3", "vf_w1", "vf_b1", "vf_w2", "vf_b2", "vf_w3", "vf_b3", "log_std")


def param_shapes(obs_dim, act_dim, hidden=64):
    return {"pi_w1": (hidden, obs_dim), "pi_b1": (hidden,), "pi_w2": (hidden, hidden), "pi_b2": (hidden,),
            "pi_w3": (act_dim, hidden), "pi_b3": (act_dim,), "vf_w1": (hidden, obs_dim), "vf_b1": (hidden,),
            "vf_w2": (hidden, hidden), "vf_b2": (hidden,), "vf_w3": (1, hidden), "vf_b3": (1,), "log_std": (act_dim,)}


def pack(params, obs_dim, act_dim, train=False):
    """params: {name: float32 array in torch layout} -> the packed blob (numpy float32); train: the PPO-gradient
    blob (the rollout blob followed by the transposed weights)."""
    lib = load()
    n = (lib.rp_train_packed_floats if train else lib.rp_packed_floats)(obs_dim, act_dim)
    if n < 0:
        check(int(n))
    shapes = param_shapes(obs_dim, act_dim)
    keep = [np.ascontiguousarray(params[k], dtype=np.float32).reshape(shapes[k]) for k in PARAM_ORDER]
    st = MlpParams(*[a.ctypes.data_as(ctypes.c_void_p) for a in keep])
    out = np.zeros(n, np.float32)
    check((lib.rp_pack_train if train else lib.rp_pack)(ctypes.byref(st), obs_dim, act_dim, out.ctypes.data_as(ctypes.c_void_p)))
    return out


def grad_layout(obs_dim, act_dim):
    """{parameter name: (offset, shape)} inside the gradient vector of rp_ppo_grad_dev, plus "pi_loss" / "vf_loss"."""
    gs = int(load().rp_grad_floats(obs_dim, act_dim)) // 2
    out = {}
    for net, n_out, base in (("pi", act_dim, 0), ("vf", 1, gs)):
        o = base
        for name, shape in (("w1", (64, obs_dim)), ("b1", (64,)), ("w2", (64, 64)), ("b2", (64,)), ("w3", (n_out, 64)), ("b3", (n_out,))):
            out["%s_%s" % (net, name)] = (o, shape)
            o += int(np.prod(shape))
        if net == "pi":
            out["log_std"] = (o, (act_dim,))
        o += n_out
        out["%s_loss" % net] = (o, ())
    return out, 2 * gs


def gather_map(obs_dim, act_dim, train=False):
    """Index map m (int64, length rp_packed_floats) with packed = concat(params in PARAM_ORDER, [0.0])[m]: the
    order depends on the dimensions only, so the device-side packing of changing parameters is one gather."""
    shapes = param_shapes(obs_dim, act_dim)
    params, off = {}, 0
    for k in PARAM_ORDER:
        size = int(np.prod(shapes[k]))
        params[k] = (np.arange(off, off + size, dtype=np.float32) + 1.0).reshape(shapes[k])     # index + 1; 0 marks padding
        off += size
    assert off < (1 << 24)                     # exact in float32
    m = pack(params, obs_dim, act_dim, train).astype(np.int64) - 1
    m[m < 0] = off                             # the appended zero
    return m, off
