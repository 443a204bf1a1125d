"""In-memory simulator host for driving `HydrodynamicsBehavior` without Isaac Sim.

Isaac Sim needs NVIDIA RTX + PhysX GPU and cannot run on an MI355X box, so
"drops into the physics-step callback" is demonstrated against this host: a
dict-backed prim / attribute store, a `RigidPrimView` look-alike over torch
tensors and a physics-step event source.  Used by tests/ and examples only.
"""
from __future__ import annotations

from typing import Callable, Sequence

import torch

from . import config as cfg
from .config import AttributeStore


class FakeWorld:
    """Rigid-body state of every prim, array-of-structs like the simulator's tensor API."""

    def __init__(self, device: str = "cuda:0"):
        self.device = torch.device(device)
        self.paths: list[str] = []
        self._index: dict[str, int] = {}
        self.positions = torch.zeros((0, 3), device=self.device)
        self.orientations = torch.zeros((0, 4), device=self.device)     # wxyz
        self.velocities = torch.zeros((0, 6), device=self.device)
        self.masses = torch.zeros((0,), device=self.device)
        self.applied = _Applied()
        self.apply_calls = 0

    def add_body(self, path: str, position, orientation_wxyz, velocity6, mass: float) -> int:
        f = lambda x, w: torch.as_tensor(x, dtype=torch.float32, device=self.device).reshape(1, w)   # noqa: E731
        self.paths.append(path)
        self.positions = torch.cat([self.positions, f(position, 3)])
        self.orientations = torch.cat([self.orientations, f(orientation_wxyz, 4)])
        self.velocities = torch.cat([self.velocities, f(velocity6, 6)])
        self.masses = torch.cat([self.masses, torch.tensor([mass], dtype=torch.float32, device=self.device)])
        return len(self.paths) - 1

    def index(self, path: str) -> int:
        if len(self._index) != len(self.paths):         # paths appended in bulk (build_c3_scene)
            self._index = {p: i for i, p in enumerate(self.paths)}
        return self._index[path]


class _Applied:
    """`world.applied[path]` -> (force, torque) the simulator last received for that prim.  Views register which
    row of which batch a prim is; rows are sliced out when somebody looks, not on every physics step."""

    def __init__(self):
        self._where: dict[str, tuple["FakeRigidView", int]] = {}

    def __getitem__(self, path: str):
        view, k = self._where[path]
        if view._force is None:
            raise KeyError(path)
        return view._force[k].clone(), view._torque[k].clone()

    def __contains__(self, path: str) -> bool:
        return path in self._where and self._where[path][0]._force is not None


class FakeRigidView:
    """The six RigidPrimView methods the plugin uses, over a FakeWorld.  Like the simulator's tensor API it hands
    out the SAME device buffers every physics step (refreshed in place from the world's state), and it consumes the
    applied wrench with two device copies."""

    def __init__(self, world: FakeWorld, prim_paths: Sequence[str], name: str, buffers: str = "stable"):
        """buffers: what `get_world_poses / get_velocities(clone=False)` hand out -
        "stable" the same device tensors every step (a tensor API's views; the best case for the plugin's prepared
        launch), "fresh" newly allocated tensors every step, "strided" non-contiguous views of wider tensors,
        "numpy" host NumPy arrays (a view created without the torch backend, hydrodynamics_behavior.py:50-51),
        "static" the same device tensors every step, filled ONCE at initialize() and never refreshed, and an apply that only
        counts: a view that costs the host nothing, so that what is left of a physics step is the plugin's own work
        (bench.py `plugin_own_us_per_step`)."""
        if buffers not in ("stable", "fresh", "strided", "numpy", "static"):
            raise ValueError(buffers)
        self.buffers = buffers
        self.world, self.name = world, name
        self.idx = torch.tensor([world.index(p) for p in prim_paths], dtype=torch.long, device=world.device)
        self.paths = list(prim_paths)
        self._ok = False
        self.fail_next_fetch = False
        n = len(self.paths)
        self._pos = torch.empty((n, 3), device=world.device)
        self._quat = torch.empty((n, 4), device=world.device)
        self._vel = torch.empty((n, 6), device=world.device)
        self._force = self._torque = None

    def initialize(self):
        self._ok = True
        if self.buffers == "static":
            torch.index_select(self.world.positions, 0, self.idx, out=self._pos)
            torch.index_select(self.world.orientations, 0, self.idx, out=self._quat)
            torch.index_select(self.world.velocities, 0, self.idx, out=self._vel)

    def is_valid(self):
        return self._ok

    def get_world_poses(self, clone=False):
        if self.fail_next_fetch:
            self.fail_next_fetch = False
            raise RuntimeError("simulated tensor API failure")
        if self.buffers == "static":
            return self._pos, self._quat
        torch.index_select(self.world.positions, 0, self.idx, out=self._pos)
        torch.index_select(self.world.orientations, 0, self.idx, out=self._quat)
        if clone or self.buffers != "stable":
            return self._hand_out(self._pos), self._hand_out(self._quat)
        return self._pos, self._quat

    def get_velocities(self, clone=False):
        if self.buffers == "static":
            return self._vel
        torch.index_select(self.world.velocities, 0, self.idx, out=self._vel)
        return self._hand_out(self._vel) if clone or self.buffers != "stable" else self._vel

    def _hand_out(self, t: torch.Tensor):
        if self.buffers == "numpy":
            return t.cpu().numpy()
        if self.buffers == "strided":                   # every row padded by one column: same values, not contiguous
            wide = torch.empty((t.shape[0], t.shape[1] + 1), device=t.device, dtype=t.dtype)
            wide[:, :t.shape[1]] = t
            return wide[:, :t.shape[1]]
        return t.clone()

    def get_masses(self, clone=False):
        m = self.world.masses[self.idx]
        return m.cpu().numpy() if self.buffers == "numpy" else m

    def apply_forces_and_torques_at_pos(self, forces=None, torques=None, positions=None, is_global=True):
        self.world.apply_calls += 1
        if self.buffers == "static":
            return
        if self._force is None or self._force.shape != forces.shape:
            self._force, self._torque = torch.empty_like(forces), torch.empty_like(torques)
            for k, p in enumerate(self.paths):
                self.world.applied._where[p] = (self, k)
        self._force.copy_(forces)
        self._torque.copy_(torques)


class FakeHost:
    """`SimHost` over a FakeWorld; `step(dt)` fires every subscribed physics-step callback
    once, in subscription order, like Kit does."""

    def __init__(self, world: FakeWorld, config_path: str | None = None, view_buffers: str = "stable"):
        self.world = world
        self.device = str(world.device)
        self._config_path = config_path
        self._subs: list[Callable[[float], None]] = []
        self.views: list[FakeRigidView] = []
        self.view_buffers = view_buffers
        self.lifecycle: list[str] = []                  # what the plugin asked of "Kit", in order (tests read it)
        self.callbacks_fired = 0

    # the two Kit calls of the reference's lifecycle besides the variable store (:50-51, :70, :126)
    def ensure_simulation_context(self):
        self.lifecycle.append("simulation_context(torch)")
        return self

    def request_property_rebuild(self):
        self.lifecycle.append("request_rebuild")

    # exposed variables
    def create_exposed_variables(self, prim: AttributeStore, variables):
        self.lifecycle.append("create_exposed_variables")
        for v in variables:
            prim.create(cfg.full_attr_name(v["attr_name"]), v["default_value"])

    def remove_exposed_variables(self, prim: AttributeStore, variables):
        self.lifecycle.append("remove_exposed_variables")
        for v in variables:
            prim.remove(cfg.full_attr_name(v["attr_name"]))
        return True

    def get_exposed_variable(self, prim: AttributeStore, full_attr_name: str) -> float:
        return prim.get(full_attr_name)

    def set_exposed_variable(self, prim: AttributeStore, full_attr_name: str, value: float) -> bool:
        return prim.set(full_attr_name, value)

    def has_rigid_body_api(self, prim: AttributeStore) -> bool:
        return prim.rigid_body

    def prim_path(self, prim: AttributeStore) -> str:
        return prim.path

    def make_rigid_view(self, prim_paths, name):
        v = FakeRigidView(self.world, prim_paths, name, self.view_buffers)
        self.views.append(v)
        return v

    def subscribe_physics_step(self, callback):
        self._subs.append(callback)
        return callback

    def unsubscribe_physics_step(self, token):
        if token in self._subs:
            self._subs.remove(token)

    def config_path(self):
        return self._config_path

    def step(self, dt: float):
        for cb in list(self._subs):
            self.callbacks_fired += 1
            cb(dt)


MAIN_SCENE = ["Obsea_Buoy", "Body"] + [f"{p}_{i}" for p in ("Coxa", "Femur", "Tibia") for i in range(6)]


def build_main_scene(batched: bool | str = True, config_path: str | None = None, seed: int = 0, device: str = "cuda:0",
                     view_buffers: str = "stable"):
    """The 20 prims of silver2_isaac_sim.usd that carry the behavior, each with its own
    `HydrodynamicsBehavior` on an in-memory host: (world, host, prims, behaviors), `on_init` done."""
    import numpy as np
    from . import behavior as hb
    rng = np.random.default_rng(seed)
    world = FakeWorld(device)
    host = FakeHost(world, config_path, view_buffers)
    prims, behaviors = [], []
    for name in MAIN_SCENE:
        buoy = name == "Obsea_Buoy"
        initial = {"xDimension": 1, "yDimension": 1, "zDimension": 3} if buoy else None
        prim = AttributeStore(name, f"/World/{'Environment' if buoy else 'SILVER2'}/{name}")
        pos = (-7, 40, 0.596) if buoy else tuple(np.array([2.0, 10.7, -18.44]) + rng.uniform(-0.3, 0.3, 3))
        q = rng.normal(0, 1, 4); q /= np.linalg.norm(q)
        vel = np.concatenate([rng.normal(0, 0.2, 3), rng.normal(0, 0.3, 3)])
        part = cfg.match_part(name, cfg.PART_TABLE)
        world.add_body(prim.path, pos, q, vel, cfg.PART_MASS.get(part, 700.0))
        b = hb.HydrodynamicsBehavior(prim, host, batched=batched)
        b.on_init()
        if initial:                                   # authored USD values for the buoy (no JSON part matches it)
            for k, v in initial.items():
                host.set_exposed_variable(prim, cfg.full_attr_name(k), v)
        prims.append(prim); behaviors.append(b)
    return world, host, prims, behaviors


def build_c3_scene(envs: int = 1024, batched: bool | str = True, seed: int = 3, device: str = "cuda:0"):
    """BASELINE config 3 through the PLUGIN: `envs` SILVER2 robots (body + 6 coxae + 6 femora + 6 tibiae = 19 prims each;
    19 456 prims at 1 024 envs), every prim with its own `HydrodynamicsBehavior`, parameters from the JSON part table,
    masses as in silver2_isaac_sim.usd.  (world, host, prims, behaviors), `on_init` done."""
    import numpy as np
    from . import behavior as hb
    from . import scenes
    sc = scenes.scene_c3(envs=envs, seed=seed)
    world = FakeWorld(device)
    dev = world.device
    st = sc.state
    world.positions = torch.from_numpy(np.ascontiguousarray(st[:, 0:3])).to(dev)
    world.orientations = torch.from_numpy(np.ascontiguousarray(st[:, [6, 3, 4, 5]])).to(dev)
    world.velocities = torch.from_numpy(np.ascontiguousarray(st[:, 7:13])).to(dev)
    world.masses = torch.from_numpy(np.ascontiguousarray(sc.params[:, 10])).to(dev)
    host = FakeHost(world)
    links = ["Body"] + [f"{p}_{i}" for p in ("Coxa", "Femur", "Tibia") for i in range(6)]
    prims, behaviors = [], []
    for e in range(envs):
        for name in links:
            prim = AttributeStore(name, f"/World/envs/env_{e}/SILVER2/{name}")
            world.paths.append(prim.path)
            b = hb.HydrodynamicsBehavior(prim, host, batched=batched)
            b.on_init()
            prims.append(prim); behaviors.append(b)
    return world, host, prims, behaviors, sc
