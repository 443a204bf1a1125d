"""Parameter schema and JSON override rule of the hydrodynamics plugin.

Mirrors the reference's configuration surface so a scene authored for it works
unchanged:

  * the 12 exposed USD attributes `exposedVar:hydrodynamicsBehavior:<name>`
    with their types and defaults  (hydrodynamics_behavior.py:28-46,68,115);
  * the two-level override applied at `on_init`: `globals` first, then the
    first `parts` key (dict order) that is a substring of the lower-cased prim
    name, with "body" as the fall-back key  (hydrodynamics_behavior.py:72-112);
  * the per-part values shipped with the reference
    (hydrodynamics_config.json:2-54) and the `physics:mass` values of the
    SILVER2 links (silver2_isaac_sim.usd).
"""
from __future__ import annotations

import json
import logging
import os
from typing import Any, Mapping

log = logging.getLogger("silver2_isaacsim_amd")

EXPOSED_ATTR_NS = "exposedVar"          # isaacsim.replicator.behavior.global_variables.EXPOSED_ATTR_NS
BEHAVIOR_NS = "hydrodynamicsBehavior"   # hydrodynamics_behavior.py:26

# (name, default, doc) in the reference's order - hydrodynamics_behavior.py:28-46
_SCHEMA = (
    ("waterDensity", 1025.0, "Density of the fluid in kg/m^3."),
    ("gravity", 9.81, "Gravitational acceleration in m/s^2."),
    ("xDimension", 1.0, "Object dimension along its local X-axis (m)."),
    ("yDimension", 1.0, "Object dimension along its local Y-axis (m)."),
    ("zDimension", 1.0, "Object dimension along its local Z-axis (m)."),
    ("linearDragCoefficient", 1.2, "Quadratic linear drag coefficient (Cd)."),
    ("angularDragCoefficient", 0.8, "Quadratic angular drag coefficient."),
    ("linearDamping", 300.0, "Linear damping multiplier for low-speed stability."),
    ("angularDamping", 150.0, "Linear angular damping for low-speed stability."),
    ("linearAddedMassCoefficient", 0.05, "Added mass coefficient for surge, sway, and heave acceleration."),
    ("angularAddedMassCoefficient", 0.02, "Added mass coefficient for roll, pitch, and yaw acceleration."),
    ("liftCoefficient", 1.0, "A multiplier for the overall strength of the lift force."),
)
SCHEMA_NAMES = tuple(n for n, _, _ in _SCHEMA)
SCHEMA_DEFAULTS = {n: d for n, d, _ in _SCHEMA}


def variables_to_expose(float_type: Any = "float") -> list[dict]:
    """The VARIABLES_TO_EXPOSE list; `float_type` is Sdf.ValueTypeNames.Float under Kit."""
    return [{"attr_name": n, "attr_type": float_type, "default_value": d, "doc": doc}
            for n, d, doc in _SCHEMA]


def full_attr_name(name: str) -> str:
    return f"{EXPOSED_ATTR_NS}:{BEHAVIOR_NS}:{name}"


# hydrodynamics_config.json:2-54 (key order of `parts` matters for the match rule)
GLOBALS = {"waterDensity": 1025.0, "gravity": 9.81}


def _part(x, y, z, cd, ca, dl, da, lift, aml, ama):
    return {"xDimension": x, "yDimension": y, "zDimension": z,
            "linearDragCoefficient": cd, "angularDragCoefficient": ca,
            "linearDamping": dl, "angularDamping": da, "liftCoefficient": lift,
            "linearAddedMassCoefficient": aml, "angularAddedMassCoefficient": ama}


PART_TABLE = {
    "body":  _part(0.26, 0.26, 0.30, 1.2, 0.8, 300.0, 150.0, 0.5, 0.2, 0.1),
    "coxa":  _part(0.06, 0.06, 0.09, 0.8, 0.1, 10.0, 1.0, 0.1, 0.0, 0.0),
    "femur": _part(0.06, 0.09, 0.06, 0.9, 0.1, 15.0, 2.0, 0.1, 0.0, 0.0),
    "tibia": _part(0.06, 0.09, 0.06, 1.0, 0.1, 20.0, 2.0, 0.1, 0.0, 0.0),
}
# physics:mass of /World/SILVER2/{Body,Coxa_*,Femur_*,Tibia_*} (kg)
PART_MASS = {"body": 18.0, "coxa": 0.45, "femur": 0.75, "tibia": 0.8}

CONFIG_FILE_NAME = "hydrodynamics_config.json"


def default_config() -> dict:
    return {"globals": dict(GLOBALS), "parts": {k: dict(v) for k, v in PART_TABLE.items()}}


def write_default_config(path: str) -> str:
    with open(path, "w") as f:
        json.dump(default_config(), f, indent=2)
    return path


def load_config(path: str | None) -> dict | None:
    """JSON at `path`, the built-in table when `path` is None, or None (with a
    warning) when the file is missing - the reference then keeps USD values."""
    if path is None:
        return default_config()
    if not os.path.exists(path):
        log.warning("[Hydro] Config missing at %s", path)
        return None
    with open(path, "r") as f:
        return json.load(f)


def match_part(prim_name: str, parts: Mapping[str, Any]) -> str | None:
    """First key of `parts` (dict order) contained in the lower-cased prim name;
    fall back to 'body' if the name contains it (hydrodynamics_behavior.py:91-101)."""
    lowered = prim_name.lower()
    for category in parts.keys():
        if category.lower() in lowered:
            return category
    if "body" in lowered:
        return "body"
    return None


def resolve_overrides(prim_name: str, data: Mapping[str, Any] | None) -> dict[str, float]:
    """Values the JSON would write over the USD attributes for this prim, in the
    order they are applied (globals, then the matched part)."""
    out: dict[str, float] = {}
    if not data:
        return out
    for k, v in data.get("globals", {}).items():
        out[k] = float(v)
    parts = data.get("parts")
    if parts is not None:
        part = match_part(prim_name, parts)
        if part is not None:
            # a 'body' fall-back that is not a key raises KeyError in the reference
            # and is swallowed by its broad except (:111-112): nothing applied.
            for k, v in parts.get(part, {}).items():
                out[k] = float(v)
        else:
            log.warning("[Hydro] Config: No matching part found for %s. Using defaults.", prim_name)
    return out


def _usd_float(value) -> float:
    """The schema's attributes are Sdf.ValueTypeNames.Float (hydrodynamics_behavior.py:30-45):
    32-bit.  Round like USD does, so that e.g. the JSON's 9.81 and an authored 9.81 compare equal."""
    import struct
    return struct.unpack("<f", struct.pack("<f", float(value)))[0]


class AttributeStore:
    """Dict-backed stand-in for a USD prim's exposed attributes, used when `pxr`
    is absent (tests, headless drivers).  Keys are full attribute names; values are
    held at float32 precision like USD `float` attributes."""

    def __init__(self, name: str, path: str | None = None, rigid_body: bool = True,
                 initial: Mapping[str, float] | None = None):
        self._name = name
        self.path = path or f"/World/{name}"
        self.rigid_body = rigid_body
        self._attrs: dict[str, float] = {}
        if initial:
            for k, v in initial.items():
                self._attrs[full_attr_name(k)] = _usd_float(v)

    def GetName(self) -> str:               # noqa: N802 (USD naming)
        return self._name

    def create(self, full_name: str, default: float) -> None:
        self._attrs.setdefault(full_name, _usd_float(default))

    def has(self, full_name: str) -> bool:
        return full_name in self._attrs

    def set(self, full_name: str, value: float) -> bool:
        if full_name not in self._attrs:
            return False
        self._attrs[full_name] = _usd_float(value)
        return True

    def get(self, full_name: str) -> float:
        return self._attrs[full_name]

    def remove(self, full_name: str) -> None:
        self._attrs.pop(full_name, None)
