"""CPU-side checks of the drop-in boundary: the C-ABI library loads and exports every symbol that
include/adm_hip.h declares (no compute calls -- there is no GPU in the build container)."""
import os
import re

from autodiffusion_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "adm_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(adm_[a-z0-9_]+)\s*\(", text)))


def test_header_and_binding_agree():
    assert _declared_symbols() == sorted(_lib.SIGNATURES)


import pytest


@pytest.mark.parametrize("kind", ["bf16", "f16"])   # libadm_hip.so and its IEEE-half build libadm_hip_f16.so
def test_library_loads_and_exports_every_symbol(kind):
    lib = _lib.load(kind)
    for name in _declared_symbols():
        assert hasattr(lib, name), name
    assert lib.adm_abi_version() == _lib.ABI_VERSION
    assert lib.adm_packed_weight_elems(192, 192, 9) == 6 * 9 * 12 * 512
    assert lib.adm_packed_weight_elems(192, 100, 9) == -1


def test_argument_errors_surface_as_exceptions_without_a_gpu():
    import ctypes as C
    import pytest
    lib = _lib.load()
    st = lib.adm_conv(None, None)
    assert st == -1 and b"null" in lib.adm_last_error()
    with pytest.raises(_lib.AdmError):
        _lib.check(st, "adm_conv")
    assert lib.adm_pack_u8_nhwc(None, None, 1, 3, 8, 8, None) == -1
    co = _lib.StepCoefs()
    assert lib.adm_ddim_step(None, None, None, None, None, None, None, 1, 3, 8, 8, C.byref(co), None) == -1
