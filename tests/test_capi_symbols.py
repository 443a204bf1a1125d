"""The C-ABI library loads on a box without a GPU and exports every function that
include/hydro.h declares; the ctypes prototypes cover exactly that set.  No compute calls."""
import os
import re

from conftest import REPO


def header_functions():
    text = open(os.path.join(REPO, "include", "hydro.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(hydro_[a-z0-9_]+)\s*\(", text)))


def test_header_declares_the_survey_minimum():
    names = header_functions()
    for need in ("hydro_create", "hydro_destroy", "hydro_set_scene", "hydro_set_params_f32", "hydro_set_params_f16",
                 "hydro_reset_prev_velocity", "hydro_step_wrench", "hydro_step_components", "hydro_kinetic_energy",
                 "hydro_sync", "hydro_last_error", "hydro_version"):
        assert need in names


def test_header_is_plain_c():
    """The boundary is a C ABI: the header must compile as C99 on its own (what a cgo / FFI binding would include)."""
    import subprocess
    r = subprocess.run(["gcc", "-fsyntax-only", "-x", "c", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror",
                        os.path.join(REPO, "include", "hydro.h")], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr


def test_library_exports_every_declared_symbol(native_built):
    from silver2_isaacsim_amd import _native
    lib = _native.load()
    names = header_functions()
    assert sorted(_native.SIGNATURES) == names
    for n in names:
        assert hasattr(lib, n), n
    assert lib.hydro_version() == 0x000701
    assert lib.hydro_status_string(0) == b"HYDRO_OK"
    assert lib.hydro_status_string(-5) == b"HYDRO_E_STATE"


def test_argument_errors_without_a_device(native_built):
    """Null / bad arguments are status codes, never crashes (no device needed to get that far)."""
    import ctypes
    from silver2_isaacsim_amd import _native
    lib = _native.load()
    assert lib.hydro_create(0, 0, None) == -1                 # HYDRO_E_ARG
    h = ctypes.c_void_p()
    assert lib.hydro_create(0, -5, ctypes.byref(h)) == -1
    assert lib.hydro_destroy(None) == -1
    assert lib.hydro_sync(None) == -1
    assert lib.hydro_capacity(None) == 0
    assert lib.hydro_last_error(None) == b"null handle"
    assert lib.hydro_set_scene(None, 1025.0, 9.81) == -1
    assert lib.hydro_set_semantics(None, 0) == -1
    assert lib.hydro_set_tuning(None, 0, 0, -1, -1) == -1


def test_missing_library_fails_loudly(tmp_path):
    import pytest
    from silver2_isaacsim_amd import _native
    with pytest.raises(OSError, match="no CPU fallback"):
        _native.load(str(tmp_path / "libhydro.so"))


def test_product_never_touches_the_oracle():
    """Nothing under silver2_isaacsim_amd/ may import, load or execute anything under oracle/
    or the test-only host emulation."""
    pkg = os.path.join(REPO, "silver2_isaacsim_amd")
    for root, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                src = open(os.path.join(root, f)).read()
                code = "\n".join(l for l in src.split("\n") if not l.strip().startswith(("#", "//", "*", "/*")))
                assert "import oracle" not in code and "from oracle" not in code, f
                assert "libhydro_oracle" not in code and "libemul" not in code, f
