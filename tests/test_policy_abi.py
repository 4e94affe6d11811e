"""The fused policy library (include/roboy_policy.h) without a GPU: it loads, exports every declared symbol, and its
host-side packing is a pure permutation (with zero padding) of the parameters that depends on the dimensions only."""
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    from gym_roboy_amd import _policy_native as pn
    lib = pn.load()
    header = open(os.path.join(ROOT, "include", "roboy_policy.h")).read()
    declared = set(re.findall(r"\b(rp_[a-z_0-9]+)\s*\(", header))
    assert declared == set(pn.SIGNATURES), declared ^ set(pn.SIGNATURES)
    for name in declared:
        assert getattr(lib, name) is not None
    assert lib.rp_abi_version() == pn.RP_ABI_VERSION


@pytest.mark.parametrize("obs_dim,act_dim", [(9, 8), (60, 38), (3, 1), (95, 64), (10, 33)])
def test_packing_is_a_permutation_with_zero_padding(obs_dim, act_dim):
    from gym_roboy_amd import _policy_native as pn
    m, total = pn.gather_map(obs_dim, act_dim)
    assert len(m) == pn.load().rp_packed_floats(obs_dim, act_dim) and len(m) % 4 == 0
    used = m[m < total]
    assert sorted(used) == list(range(total))                 # every parameter exactly once
    rng = np.random.default_rng(obs_dim)
    params = {k: rng.normal(size=s).astype(np.float32) for k, s in pn.param_shapes(obs_dim, act_dim).items()}
    flat = np.concatenate([params[k].reshape(-1) for k in pn.PARAM_ORDER] + [np.zeros(1, np.float32)])
    assert np.array_equal(flat[m], pn.pack(params, obs_dim, act_dim))


def test_unsupported_dimensions_are_refused():
    from gym_roboy_amd import _policy_native as pn
    lib = pn.load()
    assert lib.rp_packed_floats(96, 8) < 0 and lib.rp_packed_floats(9, 65) < 0 and lib.rp_packed_floats(0, 8) < 0
    assert b"supported" in lib.rp_last_error()


def _U(r):
    return (r & 3) + 8 * (r >> 2)


@pytest.mark.parametrize("obs_dim,act_dim", [(9, 8), (60, 38)])
def test_packed_blob_follows_the_documented_operand_order(obs_dim, act_dim):
    """An independent statement (numpy) of the layout rp_pack / rp_pack_train write (include/roboy_policy.h,
    csrc/mlp_common.hpp): lane l of the A operand of an MFMA holds [row l & 31][k = l >> 5]; a hidden layer's K pair of
    register r is the units (U(r), U(r) + 4) of a 32-row tile."""
    from gym_roboy_amd import _policy_native as pn
    rng = np.random.default_rng(1)
    p = {k: rng.normal(size=s).astype(np.float32) for k, s in pn.param_shapes(obs_dim, act_dim).items()}
    blob = pn.pack(p, obs_dim, act_dim, train=True)
    k1s, ot_pi = (obs_dim + 2) // 2, (act_dim + 31) // 32
    o = 0
    lanes = np.arange(64)
    row, hf = lanes & 31, lanes >> 5
    for net in ("pi", "vf"):                                   # layer 1: 4 row tiles, k = column of [W1 | b1 | 0]
        w1b = np.concatenate([p[net + "_w1"], p[net + "_b1"][:, None], np.zeros((64, 2 * k1s - obs_dim - 1), np.float32)], axis=1)
        for m in range(2):
            for s in range(k1s):
                assert np.array_equal(blob[o:o + 64], w1b[32 * m + row, 2 * s + hf]), (net, m, s)
                o += 64
    t3 = {}
    for net, n_out, ot in (("pi", act_dim, ot_pi), ("vf", 1, 1)):
        for oo in range(2):
            for m in range(2):
                for r in range(16):
                    assert np.array_equal(blob[o:o + 64], p[net + "_w2"][32 * oo + row, 32 * m + _U(r) + 4 * hf])
                    o += 64
        for oo in range(2):
            assert np.array_equal(blob[o:o + 64], np.where(hf == 0, p[net + "_b2"][32 * oo + row], 0.0).astype(np.float32))
            o += 64
        w3 = np.zeros((32 * ot, 64), np.float32); w3[:n_out] = p[net + "_w3"]
        b3 = np.zeros(32 * ot, np.float32); b3[:n_out] = p[net + "_b3"]
        for q in range(ot):
            for m in range(2):
                for r in range(16):
                    assert np.array_equal(blob[o:o + 64], w3[32 * q + row, 32 * m + _U(r) + 4 * hf])
                    o += 64
        for q in range(ot):
            assert np.array_equal(blob[o:o + 64], np.where(hf == 0, b3[32 * q + row], 0.0).astype(np.float32))
            o += 64
        t3[net] = w3
    assert np.array_equal(blob[o:o + act_dim], p["log_std"])
    o = (o + 64 + 3) & ~3
    assert o == pn.load().rp_packed_floats(obs_dim, act_dim)
    for net, n_out in (("pi", act_dim), ("vf", 1)):           # the transposed operands of the gradient kernels
        k3s = (n_out + 1) // 2
        w3p = np.zeros((2 * k3s, 64), np.float32); w3p[:n_out] = p[net + "_w3"]
        for m in range(2):
            for s in range(k3s):
                assert np.array_equal(blob[o:o + 64], w3p[2 * s + hf, 32 * m + row])
                o += 64
        for ip in range(2):
            for oo in range(2):
                for r in range(16):
                    assert np.array_equal(blob[o:o + 64], p[net + "_w2"][32 * oo + _U(r) + 4 * hf, 32 * ip + row])
                    o += 64
    assert ((o + 3) & ~3) == pn.load().rp_train_packed_floats(obs_dim, act_dim)


def test_lds_opt_in_is_remembered_per_kernel_and_per_device():
    """More than 64 KB of dynamic LDS is granted per kernel with hipFuncSetAttribute - per device: a second device in
    the same process must get its own grant (the bookkeeping behind csrc/mlp_common.hpp: grant_lds)."""
    from gym_roboy_amd import _policy_native as pn
    lib = pn.load()
    need = lib.rp_debug_lds_grant_needed
    big = 100 * 1024
    assert need(7, 40, 32 * 1024) == 0                    # under 64 KB: nothing to grant
    assert need(7, 40, big) == 1 and need(7, 40, big) == 0
    assert need(7, 41, big) == 1                          # another device: its own grant
    assert need(6, 40, big) == 1                          # another kernel too
    assert need(7, 40, big + 4096) == 1 and need(7, 40, big) == 0      # a larger request is granted again, a smaller one is covered


def test_host_permutation_is_a_keyed_bijection():
    import ctypes as c
    from gym_roboy_amd import _policy_native as pn
    lib = pn.load()
    seen = []
    for n in (1, 2, 3, 5, 64, 1000, 4097, 65536, 250000):
        out = np.empty(n, np.int64)
        assert lib.rp_perm_host(1234, n, 0, n, out.ctypes.data_as(c.c_void_p)) == 0
        assert np.array_equal(np.sort(out), np.arange(n))
        if n >= 1000:
            assert np.mean(out == np.arange(n)) < 0.01                 # not the identity
            other = np.empty(n, np.int64)
            assert lib.rp_perm_host(1235, n, 0, n, other.ctypes.data_as(c.c_void_p)) == 0
            assert np.mean(out == other) < 0.01                        # another key, another order
            # no structure a shuffle should not have: neighbours in the order are not neighbours in the batch
            assert np.mean(np.abs(np.diff(out)) == 1) < 0.01
    assert lib.rp_perm_host(1, 10, 5, 6, np.empty(6, np.int64).ctypes.data_as(c.c_void_p)) != 0
