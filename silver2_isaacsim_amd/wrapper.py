"""`HipHydrodynamicsWrapper`: the calculator surface of the reference, on the HIP engine.

Same constructor keywords and the same `calculate_hydrodynamic_forces(...)` method
as the reference's two calculators

    NumbaHydrodynamicsWrapper  numba_hydrodynamics_wrapper.py:9-53
    WarpHydrodynamicsWrapper   warp_hydrodynamics_wrapper.py:10-132

but every constructor argument may be a scalar (one body, or broadcast) or a
length-N array, and one call evaluates all N bodies in one kernel launch.

Return convention follows the Warp wrapper, which is what the behavior script
consumes (hydrodynamics_behavior.py:205-209): eight float32 device tensors of
shape (N,3) that are views of wrapper-owned buffers, valid until the next call
(warp_hydrodynamics_wrapper.py:123-132).  Values follow the Numba path where the
two references disagree (SURVEY.md notes N3, N4, N6) unless `semantics="warp"` is
passed, which reproduces the Warp twin's added-mass rotation (N3) and dry-body
centres (N6) instead - PARITY UNPINNED for that mode (restated from source text;
the reference holds no outputs of its Warp calculator).  The ninth value of the Numba tuple, the submersion ratio,
is kept in `self.sub_ratio`.
"""
from __future__ import annotations

import numpy as np
import torch

from .engine import HydroEngine

_CTOR_ORDER = ("width", "depth", "height", "linear_drag_coefficient", "angular_drag_coefficient",
               "linear_damping", "angular_damping", "water_density", "gravity",
               "linear_mass_coeff", "angular_mass_coeff", "lift_coefficient")


def _column(x, n: int, name: str) -> np.ndarray:
    a = np.asarray(x.detach().cpu() if torch.is_tensor(x) else x, dtype=np.float64).reshape(-1)
    if a.size == 1:
        return np.full(n, float(a[0]))
    if a.size != n:
        raise ValueError(f"{name}: expected a scalar or {n} values, got {a.size}")
    return a


class HipHydrodynamicsWrapper:
    def __init__(self, width, depth, height, linear_drag_coefficient, angular_drag_coefficient, linear_damping,
                 angular_damping, water_density, gravity, linear_mass_coeff, angular_mass_coeff, lift_coefficient,
                 device="cuda:0", mass=None, coeff_dtype: str = "f32", semantics: str = "numba"):
        args = dict(zip(_CTOR_ORDER, (width, depth, height, linear_drag_coefficient, angular_drag_coefficient,
                                      linear_damping, angular_damping, water_density, gravity,
                                      linear_mass_coeff, angular_mass_coeff, lift_coefficient)))
        sizes = [np.asarray(v.detach().cpu() if torch.is_tensor(v) else v).size for v in args.values()]
        if mass is not None:
            sizes.append(np.asarray(mass.detach().cpu() if torch.is_tensor(mass) else mass).size)
        n = max(sizes)
        rho = _column(water_density, n, "water_density")
        g = _column(gravity, n, "gravity")
        if np.any(rho != rho[0]) or np.any(g != g[0]):
            raise ValueError("water_density and gravity are scene scalars: one value per wrapper")
        self.device = torch.device(device)
        self.n = n
        # attributes of the reference wrappers (numba_hydrodynamics_wrapper.py:12-24)
        self.width, self.depth, self.height = (_column(args[k], n, k) for k in ("width", "depth", "height"))
        self.total_volume = self.width * self.depth * self.height
        self.water_density, self.gravity = float(rho[0]), float(g[0])
        self.linear_drag_coefficient = _column(linear_drag_coefficient, n, "linear_drag_coefficient")
        self.angular_drag_coefficient = _column(angular_drag_coefficient, n, "angular_drag_coefficient")
        self.linear_damping = _column(linear_damping, n, "linear_damping")
        self.angular_damping = _column(angular_damping, n, "angular_damping")
        self.lift_coefficient = _column(lift_coefficient, n, "lift_coefficient")
        self.linear_mass_coeff = _column(linear_mass_coeff, n, "linear_mass_coeff")
        self.angular_mass_coeff = _column(angular_mass_coeff, n, "angular_mass_coeff")
        # mass only matters for the fused wrench (safety clamp); inf = clamp never engages
        self.mass = _column(mass, n, "mass") if mass is not None else np.full(n, np.float32(3.0e38))
        params = np.stack([self.width, self.depth, self.height, self.linear_drag_coefficient,
                           self.angular_drag_coefficient, self.linear_damping, self.angular_damping,
                           self.lift_coefficient, self.linear_mass_coeff, self.angular_mass_coeff, self.mass], axis=0)
        self._engine = HydroEngine(n, self.device, self.water_density, self.gravity)
        self._engine.set_params(params.astype(np.float32), coeff_dtype)
        self._engine.set_semantics(semantics)
        dev = self._engine.device
        self._comps_aos = torch.empty((8, n, 3), dtype=torch.float32, device=dev)
        self._ratio = torch.empty((n,), dtype=torch.float32, device=dev)
        self._force = torch.empty((n, 3), dtype=torch.float32, device=dev)
        self._torque = torch.empty((n, 3), dtype=torch.float32, device=dev)
        self.sub_ratio = self._ratio

    # ---------------------------------------------------------------- helpers
    def _rows(self, x, width: int) -> torch.Tensor:
        """Accept a torch tensor (N,w) on any device or an array-like (w,) / (N,w)."""
        t = x if torch.is_tensor(x) else torch.as_tensor(np.asarray(x, dtype=np.float32))
        t = t.to(device=self._engine.device, dtype=torch.float32)
        if t.ndim == 1:
            t = t.unsqueeze(0)
        if t.shape != (self.n, width):
            raise ValueError(f"expected shape ({self.n},{width}), got {tuple(t.shape)}")
        return t

    # ---------------------------------------------------------------- surface
    def calculate_hydrodynamic_forces(self, position, orientation_quat, linear_vel, angular_vel,
                                      linear_accel, angular_accel):
        """(buoyancy_force, drag_force, lift_force, drag_torque, added_mass_force,
        added_mass_torque, center_of_buoyancy, center_of_pressure), each (N,3) float32.
        `orientation_quat` is [x, y, z, w] as for both reference calculators."""
        ins = [self._rows(x, w).contiguous() for x, w in ((position, 3), (orientation_quat, 4), (linear_vel, 3),
                                                           (angular_vel, 3), (linear_accel, 3), (angular_accel, 3))]
        # one kernel launch: (N,3)/(N,4) tensors in, the eight (N,3) outputs straight into wrapper-owned memory
        self._engine.step_components_aos(*ins, out=self._comps_aos, ratio=self._ratio)
        return tuple(self._comps_aos[k] for k in range(8))

    def calculate_wrench(self, position, orientation_quat, linear_vel, angular_vel, delta_time: float):
        """Fused path (what the behavior script does around the calculator,
        hydrodynamics_behavior.py:196-226): returns (net_force, net_torque), (N,3) each.  The
        previous-step velocity is kept inside (zero on the first call, reset by `reset()`)."""
        p, q = self._rows(position, 3).contiguous(), self._rows(orientation_quat, 4).contiguous()
        vel = torch.cat([self._rows(linear_vel, 3), self._rows(angular_vel, 3)], dim=1).contiguous()
        self._engine.step_wrench_aos(p, q, vel, delta_time, forces=self._force, torques=self._torque, quat_xyzw=True)
        return self._force, self._torque

    def reset(self) -> None:
        self._engine.reset_prev_velocity()

    @property
    def engine(self) -> HydroEngine:
        return self._engine

    def close(self) -> None:
        self._engine.close()
