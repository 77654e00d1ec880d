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
    # the CU-partition table is host state: registering / forgetting a stream's CU budget needs no GPU
    fake = C.c_void_p(0x1000)
    assert lib.adm_stream_set_cus(fake, 96) == 0 and lib.adm_stream_set_cus(fake, 64) == 0 and lib.adm_stream_set_cus(fake, 0) == 0
    assert lib.adm_stream_set_cus(fake, -1) == -1 and b"ncu" in lib.adm_last_error()
    out = C.c_void_p()
    assert lib.adm_stream_create_cumask(None, 8, C.byref(out)) == -1
    empty = (C.c_uint32 * 8)()
    assert lib.adm_stream_create_cumask(empty, 8, C.byref(out)) == -1 and b"empty mask" in lib.adm_last_error()


def test_header_is_plain_c_and_links_from_a_c_client(tmp_path):
    """include/adm_hip.h compiles as C99 (-pedantic: no C++isms, no torch types) and a C program links against the library."""
    import shutil
    import subprocess
    if shutil.which("gcc") is None:
        pytest.skip("no gcc")
    src = tmp_path / "client.c"
    src.write_text('#include "adm_hip.h"\n'
                   "int main(void) {\n"
                   "  adm_conv_args a; adm_conv2d_args b; adm_step_coefs c; (void)a; (void)b; (void)c;\n"
                   "  if (adm_conv(0, 0) != ADM_E_ARG) return 2;            /* argument errors need no GPU */\n"
                   "  if (!adm_last_error() || !adm_last_error()[0]) return 3;\n"
                   "  return adm_abi_version() == ADM_ABI_VERSION ? 0 : 1;\n"
                   "}\n")
    libdir = os.path.join(ROOT, "autodiffusion_amd")
    exe = str(tmp_path / "client")
    r = subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-I", os.path.join(ROOT, "include"), str(src), "-o", exe,
                        "-L", libdir, "-l:libadm_hip.so", "-Wl,-rpath," + libdir], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert subprocess.run([exe]).returncode == 0
