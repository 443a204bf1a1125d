"""USD-crate parameter reader.

The decoders are tested on hand-made data; the full reader is run on the reference's scene files
when they are present (build container) and must reproduce the table committed in
tests/golden/usd_hydrodynamics_tables.json and the values listed in SURVEY.md's appendix."""
import json
import os
import struct

import numpy as np
import pytest

from conftest import GOLDEN
from silver2_isaacsim_amd import usd_crate as uc

SCENES = "/root/reference/src/scenes"
needs_reference = pytest.mark.skipif(not os.path.isdir(SCENES), reason="reference scenes not available on this box")


def test_lz4_block_decoder():
    # literals only
    assert uc.lz4_block_decompress(bytes([0x50]) + b"hello", 5) == b"hello"
    # 4 literals "abcd" + match (offset 4, length 4+4=8) -> "abcd" * 3, then final literal-only sequence "!"
    blk = bytes([0x44]) + b"abcd" + struct.pack("<H", 4) + bytes([0x10]) + b"!"
    assert uc.lz4_block_decompress(blk, 13) == b"abcdabcdabcd!"
    # overlapping run: one literal 'x' then match offset 1 length 19 (15 + extension byte 0, +4)
    blk = bytes([0x1F]) + b"x" + struct.pack("<H", 1) + bytes([0]) + bytes([0x00])
    assert uc.lz4_block_decompress(blk, 20) == b"x" * 20
    with pytest.raises(uc.CrateError):
        uc.lz4_block_decompress(bytes([0x50]) + b"hello", 6)
    assert uc.fast_decompress(bytes([0]) + bytes([0x50]) + b"hello", 5) == b"hello"


def test_integer_decoder():
    # values 10, 11, 12, 112, 12, 70012 -> deltas 10, 1, 1, 100, -100, 70000; common delta 1
    codes = [1, 0, 0, 1, 1, 3]
    packed = bytearray(2)
    for i, c in enumerate(codes):
        packed[i >> 2] |= c << ((i & 3) * 2)
    buf = struct.pack("<i", 1) + bytes(packed) + struct.pack("<b", 10) + struct.pack("<b", 100) + struct.pack("<b", -100) + struct.pack("<i", 70000)
    assert uc.decode_ints(buf, 6) == [10, 11, 12, 112, 12, 70012]
    assert uc.decode_ints(b"", 0) == []


@needs_reference
@pytest.mark.parametrize("scene,n_prims,rate", [("silver2_isaac_sim.usd", 20, 60), ("silver2_isaac_sim_locomotion.usd", 19, 120),
                                               ("silver2_isaac_sim_terrestrial_environment.usd", 0, 120)])
def test_scene_tables(scene, n_prims, rate):
    tab = uc.hydrodynamics_table(os.path.join(SCENES, scene))
    assert tab["__scene__"]["timeStepsPerSecond"] == rate
    assert len(tab) - 1 == n_prims
    golden = json.load(open(os.path.join(GOLDEN, "usd_hydrodynamics_tables.json")))[scene]
    assert json.loads(json.dumps(tab)) == golden


@needs_reference
def test_main_scene_values_match_the_survey_appendix():
    tab = uc.hydrodynamics_table(os.path.join(SCENES, "silver2_isaac_sim.usd"))
    buoy = tab["/World/Environment/Obsea_Buoy"]
    f32 = lambda x: float(np.float32(x))                                    # noqa: E731
    assert (buoy["xDimension"], buoy["yDimension"], buoy["zDimension"]) == (1.0, 1.0, 3.0)
    assert buoy["linearDamping"] == 300.0 and buoy["liftCoefficient"] == 1.0 and "mass" not in buoy
    assert buoy["translate"] == pytest.approx((-7.0, 40.0, 0.596), abs=1e-3)
    body = tab["/World/SILVER2/Body"]
    assert body["xDimension"] == f32(0.259) and body["zDimension"] == f32(0.3) and body["mass"] == 18.0
    assert body["linearAddedMassCoefficient"] == f32(0.2) and body["liftCoefficient"] == 0.5
    for i in range(6):
        assert tab[f"/World/SILVER2/Coxa_{i}"]["mass"] == f32(0.45) and tab[f"/World/SILVER2/Coxa_{i}"]["linearDamping"] == 10.0
        assert tab[f"/World/SILVER2/Femur_{i}"]["mass"] == f32(0.75) and tab[f"/World/SILVER2/Femur_{i}"]["yDimension"] == f32(0.09)
        assert tab[f"/World/SILVER2/Tibia_{i}"]["mass"] == f32(0.8) and tab[f"/World/SILVER2/Tibia_{i}"]["linearDragCoefficient"] == 1.0
    assert all(t["waterDensity"] == 1025.0 and t["gravity"] == f32(9.81) for p, t in tab.items() if p != "__scene__")
    prims, rows, rho, g = uc.params_rows(tab)
    assert rows.shape == (20, 11) and rho == 1025.0 and g == f32(9.81)
    assert rows[prims.index("/World/SILVER2/Tibia_3")].tolist() == [f32(x) for x in (0.06, 0.09, 0.06, 1.0, 0.1, 20.0, 2.0, 0.1, 0.0, 0.0, 0.8)]


def test_table_from_the_committed_fixture_feeds_the_engine_rows():
    """Without the scene files: the committed table still turns into engine parameter rows."""
    golden = json.load(open(os.path.join(GOLDEN, "usd_hydrodynamics_tables.json")))["silver2_isaac_sim_locomotion.usd"]
    prims, rows, rho, g = uc.params_rows(golden)
    assert len(prims) == 19 and rows.dtype == np.float32 and rows[prims.index("/World/SILVER2/Body"), 10] == 18.0


def test_rejects_non_crate_files(tmp_path):
    p = tmp_path / "x.usd"
    p.write_bytes(b"#usda 1.0\n")
    with pytest.raises(uc.CrateError):
        uc.CrateFile(str(p))
