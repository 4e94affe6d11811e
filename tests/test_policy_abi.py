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
