"""Host logic: parameter schema + JSON override rule (hydrodynamics_behavior.py:28-46,72-112),
synthetic scene laws and the block partition (8e)."""
import json

import numpy as np
import pytest

from silver2_isaacsim_amd import config as cfg
from silver2_isaacsim_amd import scenes
from silver2_isaacsim_amd.distributed import shard_range


def test_schema_names_order_and_defaults():
    assert cfg.BEHAVIOR_NS == "hydrodynamicsBehavior" and cfg.EXPOSED_ATTR_NS == "exposedVar"
    assert cfg.SCHEMA_NAMES == ("waterDensity", "gravity", "xDimension", "yDimension", "zDimension",
                                "linearDragCoefficient", "angularDragCoefficient", "linearDamping", "angularDamping",
                                "linearAddedMassCoefficient", "angularAddedMassCoefficient", "liftCoefficient")
    d = cfg.SCHEMA_DEFAULTS
    assert (d["waterDensity"], d["gravity"], d["xDimension"], d["linearDragCoefficient"], d["angularDragCoefficient"],
            d["linearDamping"], d["angularDamping"], d["linearAddedMassCoefficient"],
            d["angularAddedMassCoefficient"], d["liftCoefficient"]) == (1025.0, 9.81, 1.0, 1.2, 0.8, 300.0, 150.0, 0.05, 0.02, 1.0)
    v = cfg.variables_to_expose("FLOAT")
    assert len(v) == 12 and all(set(x) == {"attr_name", "attr_type", "default_value", "doc"} for x in v)
    assert cfg.full_attr_name("gravity") == "exposedVar:hydrodynamicsBehavior:gravity"


@pytest.mark.parametrize("prim,part", [("Body", "body"), ("Coxa_0", "coxa"), ("Femur_5", "femur"), ("Tibia_3", "tibia"),
                                       ("Obsea_Buoy", None), ("my_BODY_shell", "body"), ("coxa_femur", "coxa")])
def test_part_matching_rule(prim, part):
    assert cfg.match_part(prim, cfg.PART_TABLE) == part


def test_overrides_apply_globals_then_part():
    o = cfg.resolve_overrides("Tibia_2", cfg.default_config())
    assert list(o)[:2] == ["waterDensity", "gravity"]
    assert o["linearDamping"] == 20.0 and o["xDimension"] == 0.06 and o["yDimension"] == 0.09
    o = cfg.resolve_overrides("Obsea_Buoy", cfg.default_config())
    assert set(o) == {"waterDensity", "gravity"}                      # no part matched: USD values stay
    assert cfg.resolve_overrides("anything", None) == {}


def test_config_file_roundtrip_and_missing(tmp_path):
    p = cfg.write_default_config(str(tmp_path / cfg.CONFIG_FILE_NAME))
    data = cfg.load_config(p)
    assert list(data["parts"]) == ["body", "coxa", "femur", "tibia"]            # dict order drives the match
    assert data["parts"]["body"]["liftCoefficient"] == 0.5 and data["globals"]["waterDensity"] == 1025.0
    assert cfg.load_config(str(tmp_path / "nope.json")) is None
    data["parts"]["body"]["linearDamping"] = 123.0
    json.dump(data, open(p, "w"))
    assert cfg.resolve_overrides("Body", cfg.load_config(p))["linearDamping"] == 123.0


def test_attribute_store():
    prim = cfg.AttributeStore("Coxa_1")
    name = cfg.full_attr_name("gravity")
    assert not prim.set(name, 1.0)                      # not created yet -> refused, like an invalid USD attr
    prim.create(name, 9.81)
    f32 = lambda x: float(np.float32(x))                  # noqa: E731  (USD float attributes are 32-bit)
    assert prim.get(name) == f32(9.81) and prim.set(name, 3.7) and prim.get(name) == f32(3.7)
    prim.create(name, 9.81)                             # create never clobbers an authored value
    assert prim.get(name) == f32(3.7) and prim.GetName() == "Coxa_1" and prim.path == "/World/Coxa_1"


# ---------------------------------------------------------------- scenes
def test_scene_shapes_dtypes_and_determinism():
    for name, n in (("c1", 1), ("c2", 4096), ("c3", 19456)):
        a, b = scenes.make_scene(name), scenes.make_scene(name)
        assert a.n == n and a.state.shape == (n, 13) and a.prev.shape == (n, 6) and a.params.shape == (n, 11)
        assert a.state.dtype == a.prev.dtype == a.params.dtype == np.float32
        assert np.array_equal(a.state, b.state) and np.array_equal(a.params, b.params)
    assert scenes.scene_c3().dt == float(np.float32(1 / 120)) and scenes.scene_c2().dt == float(np.float32(1 / 60))


def test_c3_uses_the_shipped_link_parameters():
    sc = scenes.scene_c3(envs=4)
    assert sc.n == 76
    body, coxa, femur, tibia = sc.params[0], sc.params[1], sc.params[7], sc.params[13]
    assert np.allclose(body, [0.26, 0.26, 0.30, 1.2, 0.8, 300, 150, 0.5, 0.2, 0.1, 18.0])
    assert np.allclose(coxa, [0.06, 0.06, 0.09, 0.8, 0.1, 10, 1, 0.1, 0, 0, 0.45])
    assert np.allclose(femur, [0.06, 0.09, 0.06, 0.9, 0.1, 15, 2, 0.1, 0, 0, 0.75])
    assert np.allclose(tibia, [0.06, 0.09, 0.06, 1.0, 0.1, 20, 2, 0.1, 0, 0, 0.8])
    assert np.all(sc.state[:, 2] < -17.0)                                     # robot is deep under water


def test_c4_population_mix_and_margin_rule():
    sc = scenes.scene_c4(n=16384, seed=99)
    ext = scenes.vertical_extent(sc.state[:, 3:7], sc.params[:, :3])
    pz = sc.state[:, 2].astype(np.float64)
    dry, full = (pz - ext >= 0).mean(), (pz + ext <= 0).mean()
    assert 0.22 < dry < 0.28 and 0.47 < full < 0.53
    speed = np.linalg.norm(sc.state[:, 7:10], axis=1)
    assert 0.005 < (speed == 0).mean() < 0.015 and 0.03 < ((speed > 0) & (speed < 0.2)).mean() < 0.07
    assert scenes.branch_margins(sc.state, sc.params).min() >= 1e-4
    raw = scenes.scene_c4(n=16384, seed=99, margin=None)
    assert scenes.branch_margins(raw.state, raw.params).min() < 1e-4 and sc.info["resampled"] > 0


def test_c5_coefficients_are_fp16_representable():
    sc = scenes.scene_c5(n=4096)
    co = sc.params[:, 3:10]
    assert sc.coeff_dtype == "f16" and np.array_equal(co.astype(np.float16).astype(np.float32), co)


@pytest.mark.parametrize("n,world", [(262144, 8), (10, 3), (7, 8), (0, 2), (1048576, 6)])
def test_shard_ranges_partition_the_bodies(n, world):
    ranges = [shard_range(n, r, world) for r in range(world)]
    assert ranges[0][0] == 0 and ranges[-1][1] == n
    assert all(ranges[i][1] == ranges[i + 1][0] for i in range(world - 1))
    sizes = [hi - lo for lo, hi in ranges]
    assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        shard_range(n, world, world)


def test_scene_shard_is_a_view_of_the_block():
    sc = scenes.scene_c2()
    parts = [sc.shard(r, 4) for r in range(4)]
    assert np.array_equal(np.concatenate([p.state for p in parts]), sc.state)
    assert parts[2].info["shard"][:2] == (2, 4)


def test_kit_host_follows_the_reference_rule_for_a_missing_json(tmp_path, caplog):
    """hydrodynamics_behavior.py:76-81: the JSON beside the script is applied; without one, a warning and the USD values
    stay.  KitHost hands load_config the path either way (the shipped table only when use_builtin_table=True)."""
    import json
    import logging
    from silver2_isaacsim_amd import config as cfg
    from silver2_isaacsim_amd.behavior import KitHost
    empty = tmp_path / "no_json_here"; empty.mkdir()
    host = KitHost(config_dir=str(empty))
    p = host.config_path()
    assert p == str(empty / cfg.CONFIG_FILE_NAME)
    with caplog.at_level(logging.WARNING):
        assert cfg.load_config(p) is None                     # -> _apply_json_config returns: USD values untouched
    assert "Config missing" in caplog.text
    assert KitHost(config_dir=str(empty), use_builtin_table=True).config_path() is None       # load_config(None) = the shipped table
    assert cfg.load_config(None) == cfg.default_config()
    edited = dict(cfg.default_config()); edited["globals"] = {"waterDensity": 999.0, "gravity": 9.8}
    (empty / cfg.CONFIG_FILE_NAME).write_text(json.dumps(edited))
    for h in (KitHost(config_dir=str(empty)), KitHost(config_dir=str(empty), use_builtin_table=True)):
        assert cfg.load_config(h.config_path())["globals"]["waterDensity"] == 999.0      # a user's JSON always wins


def test_default_kit_host_applies_the_shipped_table_like_the_reference(tmp_path):
    """The reference ships hydrodynamics_config.json beside hydrodynamics_behavior.py and applies it at on_init (:72-112).
    A default drop-in - KitHost() with no config_dir - must run with the same parameters out of the box: it resolves to
    the table the package carries as constants (config.default_config() = hydrodynamics_config.json:2-54).  A host that
    names a directory keeps the literal rule for a missing file (warning, USD values stay)."""
    import json
    import os
    from silver2_isaacsim_amd import config as cfg
    from silver2_isaacsim_amd.behavior import KitHost
    data = cfg.load_config(KitHost().config_path())
    assert data == cfg.default_config()
    assert list(data["parts"]) == ["body", "coxa", "femur", "tibia"]          # key order is the match order (:91-101)
    tib = cfg.resolve_overrides("Tibia_3", data)                              # what a SILVER2 tibia gets out of the box
    assert tib["linearDamping"] == 20.0 and tib["yDimension"] == 0.09 and tib["waterDensity"] == 1025.0
    assert "xDimension" not in cfg.resolve_overrides("Obsea_Buoy", data)       # no part matches: USD values stay, globals applied
    # explicit directory without a JSON: the reference's rule; explicit opt-out for the default directory too
    assert KitHost(config_dir=str(tmp_path)).config_path() == str(tmp_path / cfg.CONFIG_FILE_NAME)
    assert KitHost(use_builtin_table=False).config_path().endswith(cfg.CONFIG_FILE_NAME)
    # a JSON dropped beside the package wins over the constants (a user's file always does)
    ref = "/root/reference/src/scripts/physics/hydrodynamics_config.json"
    if os.path.exists(ref):                                                   # (the build container only: the reference never travels)
        assert json.load(open(ref)) == data
