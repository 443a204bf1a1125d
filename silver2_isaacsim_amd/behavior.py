"""`HydrodynamicsBehavior`: the reference's physics-step plugin on the HIP engine.

Mirrors `HydrodynamicsBehavior(BehaviorScript)` of the reference
(src/scripts/physics/hydrodynamics_behavior.py:21-249): same class name,
`BEHAVIOR_NS`, `VARIABLES_TO_EXPOSE` (12 float attributes, same defaults), same
lifecycle (`on_init / on_play / on_stop / on_destroy / _on_physics_step`), same
JSON override rule, same error behaviour (state-fetch failures skip the step,
:191-192; a prim without RigidBodyAPI gets a warning and no view, :144-146).

What changes is underneath.  The reference creates one calculator and one
`RigidPrimView` of size 1 per prim and pays ~40 GPU launches per prim per step
(SURVEY.md 3.2).  Here every instance registers its prim with a process-wide
`EngineRegistry`, and the GROUP - not the prim - subscribes to the physics-step
event: one callback per step fetches the poses and velocities of ALL registered
prims through one batched view, runs ONE fused kernel (`hydro_step_wrench_aos`:
quaternion reorder, finite-difference acceleration, nine-component model, lever
arms, sum, clamp) and applies all wrenches with one call.  Host time per physics
step is O(1) in the number of prims (19 456 prims of config 3: one callback, not
19 456).  `batched="callbacks"` keeps one subscription per prim as the reference
does (:131-132; the first callback of a step runs the batch, the others return
after a counter comparison); `batched=False` keeps the reference's
one-view-per-prim flow (still one fused launch per prim instead of ~40).

The simulator is reached only through the small `SimHost` protocol below, so
the class runs under Kit (`KitHost`, imports omni/pxr lazily) and under the
in-memory host of `silver2_isaacsim_amd.testing` alike.
"""
from __future__ import annotations

import logging
import os
from typing import Any, Callable, Protocol, Sequence

import numpy as np
import torch

from . import config as cfg
from .engine import HydroEngine, HydroError

log = logging.getLogger("silver2_isaacsim_amd")

_STATE_FETCH_ERRORS = (UnboundLocalError, IndexError, RuntimeError, AttributeError)   # :191


class BodyView(Protocol):
    """The six `RigidPrimView` methods the plugin uses (hydrodynamics_behavior.py:139,149-153,
    172,178-179,229-234).  Tensors are torch, on the simulation device."""

    def initialize(self) -> None: ...
    def is_valid(self) -> bool: ...
    def get_world_poses(self, clone: bool = False): ...          # (N,3), (N,4) wxyz
    def get_velocities(self, clone: bool = False): ...           # (N,6) [lin | ang]
    def get_masses(self, clone: bool = False): ...               # (N,)
    def apply_forces_and_torques_at_pos(self, forces=None, torques=None, positions=None, is_global=True): ...


class SimHost(Protocol):
    """Everything the plugin needs from the simulator besides the body view."""

    device: str
    def create_exposed_variables(self, prim, variables: Sequence[dict]) -> None: ...
    def remove_exposed_variables(self, prim, variables: Sequence[dict]) -> None: ...
    def get_exposed_variable(self, prim, full_attr_name: str) -> float: ...
    def set_exposed_variable(self, prim, full_attr_name: str, value: float) -> bool: ...
    def has_rigid_body_api(self, prim) -> bool: ...
    def prim_path(self, prim) -> str: ...
    def make_rigid_view(self, prim_paths: Sequence[str], name: str) -> BodyView: ...
    def subscribe_physics_step(self, callback: Callable[[float], None]) -> Any: ...
    def config_path(self) -> str | None: ...
    # optional (looked up with getattr): what else the reference's lifecycle touches in Kit
    #   ensure_simulation_context()      SimulationContext(backend="torch", device=...)        (:50-51)
    #   request_property_rebuild()       omni.kit.window.property.get_window().request_rebuild()  (:70, :126)
    #   unsubscribe_physics_step(token)  release a subscription (Kit: dropping the token does it)


# --------------------------------------------------------------------------
# batching registry
# --------------------------------------------------------------------------
_warned_non_torch = False


def _as_device_tensor(x, dev: torch.device) -> torch.Tensor:
    """What a body view handed out -> contiguous float32 tensor on `dev`.  A view created without the torch backend
    (no `SimulationContext(backend="torch", ...)`, hydrodynamics_behavior.py:50-51) hands out NumPy arrays: they are
    converted - and copied to the device every step - with ONE warning per process, never skipped silently."""
    global _warned_non_torch
    if not torch.is_tensor(x):
        if not _warned_non_torch:
            _warned_non_torch = True
            log.warning("[Hydro] the body view returned %s, not torch tensors: is the simulation context running with "
                        "backend='torch'? Converting and copying to %s every physics step (slow, but correct).",
                        type(x).__name__, dev)
        x = torch.as_tensor(np.ascontiguousarray(x))
    return x.to(dev, torch.float32).contiguous()


class _AosStepper:
    """`hydro_step_wrench_aos` on what a body view hands out, prepared once per set of device buffers.

    A simulator's tensor API returns views of the SAME device buffers every physics step, so the launch is prepared
    once (engine.prepare_step_wrench_aos: arguments validated, ctypes values built) and re-issued while the three
    pointers stay the same; another device, dtype or layout, NumPy arrays, or new buffers go through the conversions
    and a fresh preparation (the general case: nothing pins `get_world_poses(clone=False)` to stable buffers).

    Two halves, because only the first one is a STATE FETCH whose failure skips the step (:177-192):
    `accept(...)` takes the view's tensors (conversions may raise the fetch errors), `launch(dt)` runs the kernel -
    engine errors (HYDRO_E_ARG, HYDRO_E_LAUNCH, a closed engine) propagate to the caller like any exception of the
    reference's calculator call (:205-209, outside its try)."""

    def __init__(self, engine: HydroEngine, force: torch.Tensor, torque: torch.Tensor):
        self.engine, self.force, self.torque = engine, force, torque
        self._key = None
        self._step = None
        self.prepared = 0                   # how many times the launch had to be prepared (1 for a stable-buffer host)

    def accept(self, positions, orientations, velocities) -> torch.Tensor:
        """Returns the positions tensor the kernel will read (for apply_forces_and_torques_at_pos)."""
        if torch.is_tensor(positions) and torch.is_tensor(orientations) and torch.is_tensor(velocities):
            key = (positions.data_ptr(), orientations.data_ptr(), velocities.data_ptr(), velocities.shape[0])
            if key == self._key:
                return positions
        else:
            key = None
        dev = self.engine.device
        p, q, v = (_as_device_tensor(positions, dev), _as_device_tensor(orientations, dev), _as_device_tensor(velocities, dev))
        self._step = self.engine.prepare_step_wrench_aos(p, q, v, forces=self.force, torques=self.torque)
        self.prepared += 1
        same = key is not None and key == (p.data_ptr(), q.data_ptr(), v.data_ptr(), v.shape[0])
        self._key = key if same else None               # converted copies are good for this step only
        return p

    def launch(self, dt: float) -> None:
        self._step(dt)

    def __call__(self, positions, orientations, velocities, dt: float) -> torch.Tensor:
        p = self.accept(positions, orientations, velocities)
        self.launch(dt)
        return p


class _Group:
    """All registered prims that share one host and one (water density, gravity) pair."""

    def __init__(self, host: SimHost, rho: float, g: float, semantics: str = "numba"):
        self.host, self.rho, self.g, self.semantics = host, rho, g, semantics
        self.key = None
        # registration order = row order of the batched view.  Leaving members are dropped from the set at once and from
        # the list at the next rebuild, so that on_play / on_stop of N prims is O(N), not O(N^2) (19 456 prims of config 3)
        self.members: list["HydrodynamicsBehavior"] = []
        self.member_set: set = set()
        self.callback_members = 0           # members in "callbacks" mode (their per-prim counters need levelling)
        self._rows: dict = {}               # member -> row of the current batch (the object, not its id: ids are reused)
        self.view: BodyView | None = None
        self.engine: HydroEngine | None = None
        self.dirty = True
        self.steps = 0
        self.force = self.torque = None
        self._stepper: _AosStepper | None = None
        self.batches = 0                    # physics steps for which the batch has run (see EngineRegistry.on_step)
        self.subscription = None            # the group's own physics-step subscription (scene mode)
        self.scene_members = 0              # members that rely on it
        self._engine_error_logged = False

    def add(self, b: "HydrodynamicsBehavior") -> None:
        if len(self.members) != len(self.member_set):       # somebody left since the last rebuild: compact first
            self.members = [m for m in self.members if m in self.member_set]
        self.members.append(b)
        self.member_set.add(b)
        self.dirty = True
        b._callbacks = self.batches
        if not b._scene_mode:
            self.callback_members += 1
        self._level_counters()

    def discard(self, b: "HydrodynamicsBehavior") -> None:
        self.member_set.discard(b)
        self._rows.pop(b, None)             # its previous velocity leaves with it
        self.dirty = True
        if not b._scene_mode:
            self.callback_members -= 1
        self._level_counters()

    def _level_counters(self) -> None:
        """"callbacks" mode: membership changed, everybody starts level with the batches run so far."""
        if self.callback_members:
            for m in self.members:
                m._callbacks = self.batches

    # scene mode: ONE subscription for the whole group -------------------------------------------------------
    def subscribe(self) -> Any:
        if self.subscription is None:
            self.subscription = self.host.subscribe_physics_step(self._on_physics_step)
        self.scene_members += 1
        return self.subscription

    def unsubscribe(self) -> None:
        self.scene_members -= 1
        if self.scene_members <= 0 and self.subscription is not None:
            release = getattr(self.host, "unsubscribe_physics_step", None)
            if release is not None:
                release(self.subscription)
            self.subscription = None        # under Kit dropping the token IS the unsubscribe
            self.scene_members = 0

    def _on_physics_step(self, delta_time: float) -> None:
        if delta_time <= 1e-6:              # the reference's guard (:139)
            return
        self.batches += 1
        self.step(delta_time)

    def rebuild(self) -> None:
        # Members that stay keep THEIR previous-step velocity across the rebuild (in the reference every prim owns its
        # `_last_*_velocity`, :196-198,237-238: one prim stopping does not reset the others'); a prim that joins starts
        # from zero, as after its own on_play.
        carried = None
        if self.engine is not None:
            if self.steps > 0 and self._rows:
                carried = (self.engine.get_prev_velocity(), dict(self._rows))
            self.engine.close()
        if len(self.members) != len(self.member_set):
            self.members = [m for m in self.members if m in self.member_set]
        paths = [m._prim_path for m in self.members]
        self.view = self.host.make_rigid_view(paths, "hydro_view_batched")
        self.view.initialize()
        masses = self.view.get_masses(clone=False)
        masses = (masses.detach().to("cpu", torch.float32).numpy() if torch.is_tensor(masses)
                  else np.asarray(masses, dtype=np.float32)).reshape(-1)
        rows = np.stack([m._param_row(masses[i]) for i, m in enumerate(self.members)], axis=0)
        self.engine = HydroEngine(len(paths), self.host.device, self.rho, self.g)
        self.engine.set_params(rows)
        self.engine.set_semantics(self.semantics)
        n = len(paths)
        self.force = torch.empty((n, 3), dtype=torch.float32, device=self.engine.device)
        self.torque = torch.empty((n, 3), dtype=torch.float32, device=self.engine.device)
        self._stepper = _AosStepper(self.engine, self.force, self.torque)
        self._rows = {m: i for i, m in enumerate(self.members)}
        if carried is not None:
            old_prev, old_rows = carried
            pairs = [(old_rows[m], i) for i, m in enumerate(self.members) if m in old_rows]
            if pairs:
                src = torch.tensor([p[0] for p in pairs], dtype=torch.long, device=old_prev.device)
                dst = torch.tensor([p[1] for p in pairs], dtype=torch.long, device=old_prev.device)
                prev = torch.zeros((old_prev.shape[0], n), dtype=torch.float32, device=old_prev.device)
                prev[:, dst] = old_prev[:, src]
                self.engine.set_prev_velocity(prev)
        self.dirty = False

    def step(self, dt: float) -> None:
        if self.dirty:
            self.rebuild()
        if self.view is None or not self.view.is_valid():
            return
        try:                                            # the state fetch, and only it (:177-192): failures skip the step
            positions, orientations = self.view.get_world_poses(clone=False)
            velocities = self.view.get_velocities(clone=False)
            if velocities is None or velocities.shape[0] == 0:
                return
            positions = self._stepper.accept(positions, orientations, velocities)
        except _STATE_FETCH_ERRORS:
            return
        try:
            self._stepper.launch(dt)                    # engine errors are NOT a fetch failure: they surface
        except HydroError as e:
            if not self._engine_error_logged:
                self._engine_error_logged = True
                log.error("[Hydro] the force engine refused the step for %d prims (%s); no hydrodynamic wrench is applied", len(self.member_set), e)
            raise
        self.view.apply_forces_and_torques_at_pos(forces=self.force, torques=self.torque,
                                                  positions=positions, is_global=True)
        self.steps += 1

    def close(self) -> None:
        if self.subscription is not None:
            self.scene_members = 1
            self.unsubscribe()
        if self.engine is not None:
            self.engine.close()
        self.engine = self.view = None


class EngineRegistry:
    """Process-wide: N per-prim callbacks per physics step -> one kernel launch."""

    def __init__(self):
        self._groups: dict[tuple, _Group] = {}

    def register(self, b: "HydrodynamicsBehavior") -> _Group:
        # the subscription mode is part of the key: a group is driven EITHER by its own subscription (scene mode) or by
        # its members' callbacks - mixed, a step could run the batch twice (once from a member's callback, once from the
        # group's subscription), the second time against an already-updated previous velocity
        key = (id(b._host), float(b._rho), float(b._g), b.SEMANTICS, bool(b._scene_mode))
        grp = self._groups.get(key)
        if grp is None:
            grp = self._groups[key] = _Group(b._host, b._rho, b._g, b.SEMANTICS)
            grp.key = key
        grp.add(b)
        return grp

    def unregister(self, b: "HydrodynamicsBehavior") -> None:
        grp = b._group if b._group is not None else next((g for g in self._groups.values() if b in g.member_set), None)
        if grp is None or b not in grp.member_set:
            return
        grp.discard(b)
        if not grp.member_set:
            grp.close()
            self._groups.pop(grp.key, None)

    @staticmethod
    def on_step(grp: _Group, b: "HydrodynamicsBehavior", dt: float) -> None:
        """Called once per member per physics step.  The first caller of a step runs the batch: every member counts
        its own callbacks, and the batch runs when a member's count gets ahead of the number of batches run (O(1) per
        callback; a member the simulator skips for a while simply lags and never triggers)."""
        b._callbacks += 1
        if b._callbacks > grp.batches:
            grp.batches = b._callbacks
            grp.step(dt)

    def clear(self) -> None:
        for grp in self._groups.values():
            grp.close()
        self._groups.clear()


REGISTRY = EngineRegistry()


# --------------------------------------------------------------------------
# the plugin
# --------------------------------------------------------------------------
class HydrodynamicsBehavior:
    """Drop-in counterpart of the reference's behavior script.  Under Kit, derive the
    scripted class from both this and `omni.kit.scripting.BehaviorScript`;
    `prim` / `prim_path` are then provided by Kit and `host` defaults to `KitHost()`."""

    BEHAVIOR_NS = cfg.BEHAVIOR_NS
    VARIABLES_TO_EXPOSE = cfg.variables_to_expose()
    # "numba": the documented model (numba_hydrodynamics.py).  "warp": follow warp_hydrodynamics.py - the
    # calculator the reference script instantiates (hydrodynamics_behavior.py:155) - where the two differ
    # (added-mass rotation; include/hydro.h HYDRO_SEM_WARP).  Override in the scripted subclass.
    SEMANTICS = "numba"

    def __init__(self, prim=None, host: SimHost | None = None, batched: bool | str = True):
        if prim is not None:
            self.prim = prim
        if batched not in (True, False, "scene", "callbacks"):
            raise ValueError("batched must be True / 'scene', 'callbacks' or False")
        self._host = host
        self._batched = bool(batched)
        # True / "scene": the group holds ONE physics-step subscription; "callbacks": one per prim, as the reference
        self._scene_mode = batched in (True, "scene")
        self._group: _Group | None = None
        self._engine: HydroEngine | None = None
        self._callbacks = 0                 # physics-step callbacks received since the group last changed

    # -- lifecycle -----------------------------------------------------------
    def on_init(self):
        if self._host is None:
            self._host = KitHost()
        self._device = self._host.device
        # the simulation context with the torch backend (:50-51): without it a RigidPrimView hands out NumPy arrays
        ensure = getattr(self._host, "ensure_simulation_context", None)
        self._sim_context = ensure() if ensure is not None else None
        self._hydro_calculator = None
        self._rigid_prim_view = None
        self._physx_subscription = None
        self._prim_path = self._host.prim_path(self.prim)
        self._host.create_exposed_variables(self.prim, self.VARIABLES_TO_EXPOSE)
        self._apply_json_config()
        self._request_property_rebuild()                                   # (:70)

    def _request_property_rebuild(self):
        rebuild = getattr(self._host, "request_property_rebuild", None)
        if rebuild is not None:
            rebuild()

    def _apply_json_config(self):
        """globals, then the first `parts` key contained in the lower-cased prim name
        (hydrodynamics_behavior.py:72-112)."""
        path = self._host.config_path()
        try:
            data = cfg.load_config(path)
            if data is None:
                return
            for name, value in cfg.resolve_overrides(self.prim.GetName(), data).items():
                self._set_attr(name, value)
        except Exception as e:                                    # noqa: BLE001 - reference swallows and logs (:111-112)
            log.error("[Hydro] JSON Error: %s", e)

    def _set_attr(self, name, value):
        if not self._host.set_exposed_variable(self.prim, cfg.full_attr_name(name), float(value)):
            log.warning("[Hydro] Failed to set attribute: %s", cfg.full_attr_name(name))

    def on_destroy(self):
        self._reset()
        if self._host.remove_exposed_variables(self.prim, self.VARIABLES_TO_EXPOSE) is not False:
            self._request_property_rebuild()                               # (:124-126)

    def on_play(self):
        self._setup()
        if self._batched and self._scene_mode:
            # one subscription per GROUP: host time per physics step does not grow with the number of prims
            self._physx_subscription = self._group.subscribe() if self._group is not None else None
            return
        self._physx_subscription = self._host.subscribe_physics_step(self._on_physics_step)

    def on_stop(self):
        self._reset()

    # FixedUpdate
    def _on_physics_step(self, delta_time: float):
        if delta_time <= 1e-6:
            return
        if self._batched:
            # "callbacks" mode (or a host that calls the per-prim callback itself); in scene mode the group's own
            # subscription drives the step and this method is not subscribed
            if self._group is not None and not (self._scene_mode and self._group.subscription is not None):
                EngineRegistry.on_step(self._group, self, delta_time)
            return
        if self._rigid_prim_view is None or not self._rigid_prim_view.is_valid():
            return
        self._apply_behavior(delta_time)

    def _get_exposed_variable(self, attr_name):
        return self._host.get_exposed_variable(self.prim, cfg.full_attr_name(attr_name))

    def _param_row(self, mass: float) -> np.ndarray:
        g = self._get_exposed_variable
        return np.array([g("xDimension"), g("yDimension"), g("zDimension"),
                         g("linearDragCoefficient"), g("angularDragCoefficient"),
                         g("linearDamping"), g("angularDamping"), g("liftCoefficient"),
                         g("linearAddedMassCoefficient"), g("angularAddedMassCoefficient"), mass], dtype=np.float32)

    def _setup(self):
        if not self._host.has_rigid_body_api(self.prim):
            log.warning("HydrodynamicsBehavior on prim %s requires a RigidBody component.", self._prim_path)
            return
        self._rho = self._get_exposed_variable("waterDensity")
        self._g = self._get_exposed_variable("gravity")
        if self._batched:
            self._group = REGISTRY.register(self)
            return
        name = self._prim_path.rsplit("/", 1)[-1]
        self._rigid_prim_view = self._host.make_rigid_view([self._prim_path], f"hydro_view_{name}")
        self._rigid_prim_view.initialize()
        masses = self._rigid_prim_view.get_masses(clone=False)
        self._mass = float(masses[0])
        self._engine = HydroEngine(1, self._device, self._rho, self._g)
        self._engine.set_params(self._param_row(self._mass)[None, :])
        self._engine.set_semantics(self.SEMANTICS)
        self._hydro_calculator = self._engine
        self._force = torch.empty((1, 3), dtype=torch.float32, device=self._engine.device)
        self._torque = torch.empty((1, 3), dtype=torch.float32, device=self._engine.device)
        self._stepper = _AosStepper(self._engine, self._force, self._torque)
        log.info("HydrodynamicsBehavior (HIP) initialized for %s", self._prim_path)

    def _apply_behavior(self, delta_time):
        try:
            positions, orientations = self._rigid_prim_view.get_world_poses(clone=False)
            full_velocities = self._rigid_prim_view.get_velocities(clone=False)
            if full_velocities is None or full_velocities.shape[0] == 0:
                return
            positions = self._stepper.accept(positions, orientations, full_velocities)
        except _STATE_FETCH_ERRORS:                                        # the state fetch, and only it (:177-192)
            return
        # quaternion reorder, finite-difference acceleration, model, lever arms, sum and clamp
        # (hydrodynamics_behavior.py:194-226) are one kernel; the previous velocity lives in the engine.  Engine errors
        # propagate, as an exception of the reference's calculator call (:205-209) would.
        self._stepper.launch(delta_time)
        self._rigid_prim_view.apply_forces_and_torques_at_pos(
            forces=self._force, torques=self._torque, positions=positions, is_global=True)

    def _reset(self):
        if self._group is not None:
            if self._physx_subscription is not None:
                if self._scene_mode:
                    self._group.unsubscribe()
                else:
                    release = getattr(self._host, "unsubscribe_physics_step", None)
                    if release is not None:
                        release(self._physx_subscription)
            REGISTRY.unregister(self)
            self._group = None
        elif self._physx_subscription is not None:
            release = getattr(self._host, "unsubscribe_physics_step", None)
            if release is not None:
                release(self._physx_subscription)
        if self._engine is not None:
            self._engine.close()
            self._engine = None
        self._hydro_calculator = None
        self._rigid_prim_view = None
        self._physx_subscription = None


# --------------------------------------------------------------------------
# Kit host (only importable inside Isaac Sim; nothing here runs on an AMD box)
# --------------------------------------------------------------------------
class KitHost:
    """Binds the `SimHost` protocol to Omniverse Kit / Isaac Sim.  Imports are lazy so
    that this module loads without `omni`, `pxr`, `carb` (none exist outside Kit)."""

    def __init__(self, device: str = "cuda:0", config_dir: str | None = None, use_builtin_table: bool | None = None):
        """config_dir: where hydrodynamics_config.json is looked for - the directory of the SCRIPTED behaviour file in
        the reference (hydrodynamics_behavior.py:76-77).
        use_builtin_table: with no JSON there, apply the table the reference ships (config.default_config(), the values
        of hydrodynamics_config.json:2-54) instead of the reference's rule for a missing file, which is to warn and keep
        the USD values (:79-81).  Default: True when no config_dir is given, False for an explicit one.  The reference
        always HAS its JSON beside the script, so out of the box it applies the globals and the per-part table (:76-101);
        a default `KitHost()` must run with the same physical parameters, and this package ships the table as
        constants (config.py), not as a file.  A host that names a directory gets the literal rule for it."""
        self.device = device
        self._config_dir = config_dir or os.path.dirname(os.path.abspath(__file__))
        self._use_builtin_table = (config_dir is None) if use_builtin_table is None else bool(use_builtin_table)
        self._sim_context = None

    def ensure_simulation_context(self):
        """`SimulationContext(backend="torch", device=...)` (hydrodynamics_behavior.py:50-51): the tensor API then hands
        out torch tensors on `device`; without it `RigidPrimView` returns NumPy arrays.  SimulationContext is a
        singleton in Isaac Sim, so calling this once per behavior instance is what the reference does too."""
        if self._sim_context is None:
            from omni.isaac.core.simulation_context import SimulationContext    # type: ignore
            self._sim_context = SimulationContext(backend="torch", device=self.device)
        return self._sim_context

    def request_property_rebuild(self):
        """Refresh the Property window after exposed variables appeared or went (:70, :126)."""
        import omni.kit.window.property                                      # type: ignore
        omni.kit.window.property.get_window().request_rebuild()

    def unsubscribe_physics_step(self, token):
        unsub = getattr(token, "unsubscribe", None)                          # carb subscriptions release on drop as well
        if unsub is not None:
            unsub()

    def _utils(self):
        from isaacsim.replicator.behavior.utils import behavior_utils    # type: ignore
        return behavior_utils

    def create_exposed_variables(self, prim, variables):
        from pxr import Sdf                                               # type: ignore
        typed = [dict(v, attr_type=Sdf.ValueTypeNames.Float) for v in variables]
        self._utils().create_exposed_variables(prim, cfg.EXPOSED_ATTR_NS, cfg.BEHAVIOR_NS, typed)

    def remove_exposed_variables(self, prim, variables):
        from pxr import Sdf                                               # type: ignore
        typed = [dict(v, attr_type=Sdf.ValueTypeNames.Float) for v in variables]
        if self._utils().check_if_exposed_variables_should_be_removed(prim, __file__):
            self._utils().remove_exposed_variables(prim, cfg.EXPOSED_ATTR_NS, cfg.BEHAVIOR_NS, typed)
            return True
        return False                                                          # kept: no Property-window rebuild (:124-126)

    def get_exposed_variable(self, prim, full_attr_name):
        return self._utils().get_exposed_variable(prim, full_attr_name)

    def set_exposed_variable(self, prim, full_attr_name, value):
        attr = prim.GetAttribute(full_attr_name)
        if attr and attr.IsValid():
            attr.Set(float(value))
            return True
        return False

    def has_rigid_body_api(self, prim):
        from pxr import UsdPhysics                                        # type: ignore
        return prim.HasAPI(UsdPhysics.RigidBodyAPI)

    def prim_path(self, prim):
        return str(prim.GetPath())

    def make_rigid_view(self, prim_paths, name):
        from omni.isaac.core.prims import RigidPrimView                   # type: ignore
        expr = prim_paths[0] if len(prim_paths) == 1 else list(prim_paths)
        return RigidPrimView(prim_paths_expr=expr, name=name)

    def subscribe_physics_step(self, callback):
        import omni.physx                                                 # type: ignore
        return omni.physx.get_physx_interface().subscribe_physics_step_events(callback)

    def config_path(self):
        # The reference's rule (:76-81): the JSON beside the script is applied; if there is none, a warning, and the
        # USD attribute values stay as they are (cfg.load_config does exactly that for a path that does not exist).
        # A host that ASKED for it (use_builtin_table=True), or that named no directory at all (the default deployment:
        # the reference's script always has its JSON next to it), gets the shipped table in that case (path None).
        p = os.path.join(self._config_dir, cfg.CONFIG_FILE_NAME)
        if self._use_builtin_table and not os.path.exists(p):
            return None
        return p
