"""Device-only numerics: what the CPU instantiation of csrc/hydro_body.h cannot show (it uses libm where the device uses
the fp32 hardware seeds v_rcp_f32 / v_rsq_f32 + one Newton step in fp64).  A tiny kernel in scripts/probes.hip evaluates
rcp64 / sqrt64 / rsqrt64 as the wrench kernels do; this test pins their accuracy inside the range the header promises and
their behaviour AT and BEYOND the limits it documents."""
import ctypes

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _seeds(x):
    from scripts import probes
    L = probes.lib()
    L.probe_launch_seeds.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint32, ctypes.c_void_p]
    xd = torch.from_numpy(np.asarray(x, dtype=np.float64)).to("cuda:0")
    out = torch.empty((xd.numel(), 4), dtype=torch.float64, device="cuda:0")
    assert L.probe_launch_seeds(xd.data_ptr(), out.data_ptr(), xd.numel(), None) == 0
    torch.cuda.synchronize()
    return out.cpu().numpy()


def test_seeded_reciprocal_and_roots_inside_the_promised_range(native_built):
    """1e-36 <= x <= 3e38 (the fp32 range the seeds live in): one Newton step on a 1-ulp fp32 seed squares its error -
    hydro_body.h says 1.4e-14 for 1/x and sqrt(x), 2e-14 for 1/sqrt(x)."""
    rng = np.random.default_rng(4)
    x = np.concatenate([np.exp(rng.uniform(np.log(1e-36), np.log(3e38), 200000)),
                        [1e-36, 1.0000001e-36, 1e-30, 1e-12, 1e-6, 0.2, 1.0, 4.0, 500.0, 1e6, 1e30, 3e38]])
    o = _seeds(x)
    rel = lambda got, want: np.abs(got / want - 1.0).max()      # noqa: E731
    inv_ok = x < 8.5e37                                          # 1 / x must itself be a NORMAL fp32 number: x < 2^126 (v_rcp_f32 flushes denormals)
    assert rel(o[inv_ok, 0], 1.0 / x[inv_ok]) < 2e-14
    assert np.all(o[x > 8.6e37, 0] == 0.0)                       # beyond: exactly 0, never garbage (documented in hydro_body.h)
    assert rel(o[:, 1], np.sqrt(x)) < 2e-14
    assert rel(o[:, 2], 1.0 / np.sqrt(x)) < 3e-14
    assert rel(o[:, 3], np.sqrt(x)) < 3e-14                      # |v| from the same seed as 1/|v| (solve_body)


def test_seeds_at_and_beyond_their_documented_limits(native_built):
    """What the model relies on at the edges (every use is guarded by one of its own 1e-6 thresholds):
      x = 0        sqrt64 and x * rsqrt64 are exactly 0 (0 times a FINITE seed: the seed is taken of max(x, 1e-36));
                   rsqrt64 is finite; rcp64 is not a number a caller may use (its callers guard the zero)
      0 < x < 1e-36  sqrt64 stays finite, non-negative and BELOW 1.5e-18 (it under-estimates: the seed belongs to 1e-36) -
                   anything that small is below the model's thresholds (speeds of 1e-6 squared are 1e-12)
      x > fp32 max (3.4e38)  the fp32 conversion is inf, the seeds 0: rcp64 = 0, sqrt64 = 0, rsqrt64 = 0 - a squared speed
                   of 1e39 m2/s2 is not a rigid body in water; nothing becomes NaN."""
    x = np.array([0.0, 1e-40, 1e-37, 9.99e-37, 1e39, 1e300])
    o = _seeds(x)
    assert o[0, 1] == 0.0 and o[0, 3] == 0.0 and np.isfinite(o[0, 2]) and o[0, 2] > 1e17
    assert not np.isfinite(o[0, 0]) or abs(o[0, 0]) > 1e30                   # 1/0: inf or NaN, never a plausible number
    small = o[1:4]
    assert np.isfinite(small[:, 1:]).all() and (small[:, 1] >= 0).all() and (small[:, 1] < 1.5e-18).all()
    assert (small[:, 1] <= np.sqrt(x[1:4]) * (1 + 1e-6)).all()              # (one Newton step from the seed of 1e-36: never above)
    big = o[4:]
    assert np.all(big[:, 0] == 0.0) and np.all(big[:, 1] == 0.0) and np.all(big[:, 2] == 0.0) and np.all(big[:, 3] == 0.0)
