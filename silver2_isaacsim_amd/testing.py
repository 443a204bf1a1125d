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
        self.positions = torch.zeros((0, 3), device=self.device)
        self.orientations = torch.zeros((0, 4), device=self.device)     # wxyz
        self.velocities = torch.zeros((0, 6), device=self.device)
        self.masses = torch.zeros((0,), device=self.device)
        self.applied: dict[str, tuple[torch.Tensor, torch.Tensor]] = {}
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
        return self.paths.index(path)


class FakeRigidView:
    """The six RigidPrimView methods the plugin uses, over a FakeWorld."""

    def __init__(self, world: FakeWorld, prim_paths: Sequence[str], name: str):
        self.world, self.name = world, name
        self.idx = torch.tensor([world.index(p) for p in prim_paths], dtype=torch.long, device=world.device)
        self.paths = list(prim_paths)
        self._ok = False
        self.fail_next_fetch = False

    def initialize(self):
        self._ok = True

    def is_valid(self):
        return self._ok

    def get_world_poses(self, clone=False):
        if self.fail_next_fetch:
            self.fail_next_fetch = False
            raise RuntimeError("simulated tensor API failure")
        return self.world.positions[self.idx], self.world.orientations[self.idx]

    def get_velocities(self, clone=False):
        return self.world.velocities[self.idx]

    def get_masses(self, clone=False):
        return self.world.masses[self.idx]

    def apply_forces_and_torques_at_pos(self, forces=None, torques=None, positions=None, is_global=True):
        self.world.apply_calls += 1
        for k, p in enumerate(self.paths):
            self.world.applied[p] = (forces[k].clone(), torques[k].clone())


class FakeHost:
    """`SimHost` over a FakeWorld; `step(dt)` fires every subscribed physics-step callback
    once, in subscription order, like Kit does."""

    def __init__(self, world: FakeWorld, config_path: str | None = None):
        self.world = world
        self.device = str(world.device)
        self._config_path = config_path
        self._subs: list[Callable[[float], None]] = []
        self.views: list[FakeRigidView] = []

    # exposed variables
    def create_exposed_variables(self, prim: AttributeStore, variables):
        for v in variables:
            prim.create(cfg.full_attr_name(v["attr_name"]), v["default_value"])

    def remove_exposed_variables(self, prim: AttributeStore, variables):
        for v in variables:
            prim.remove(cfg.full_attr_name(v["attr_name"]))

    def get_exposed_variable(self, prim: AttributeStore, full_attr_name: str) -> float:
        return prim.get(full_attr_name)

    def set_exposed_variable(self, prim: AttributeStore, full_attr_name: str, value: float) -> bool:
        return prim.set(full_attr_name, value)

    def has_rigid_body_api(self, prim: AttributeStore) -> bool:
        return prim.rigid_body

    def prim_path(self, prim: AttributeStore) -> str:
        return prim.path

    def make_rigid_view(self, prim_paths, name):
        v = FakeRigidView(self.world, prim_paths, name)
        self.views.append(v)
        return v

    def subscribe_physics_step(self, callback):
        self._subs.append(callback)
        return callback

    def config_path(self):
        return self._config_path

    def step(self, dt: float):
        for cb in list(self._subs):
            cb(dt)
