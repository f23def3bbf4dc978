"""CPU: the C-ABI shared library builds, loads, and exports exactly what include/gdm.h declares (no compute calls)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    text = open(os.path.join(ROOT, "include", "gdm.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(gdm_[a-z0-9_]+)\s*\(", text)))


@pytest.fixture(scope="module")
def lib_path():
    from gan_des_midi_music_gen_amd import build
    return build.build()


def test_header_and_binding_table_agree():
    from gan_des_midi_music_gen_amd import _lib
    assert _declared() == sorted(_lib.SIGNATURES), "include/gdm.h and _lib.SIGNATURES list different entry points"


def test_library_exports_every_declared_symbol(lib_path):
    lib = ctypes.CDLL(lib_path)
    for name in _declared():
        assert hasattr(lib, name), f"{name} is declared in include/gdm.h but not exported by libgdm_hip.so"


def test_probes_and_error_reporting(lib_path):
    from gan_des_midi_music_gen_amd import _lib
    lib = _lib.load()
    assert lib.gdm_arch() == b"gfx950"
    assert lib.gdm_version() >= 1
    # argument validation happens on the host before any launch: safe without a GPU
    rc = lib.gdm_gemm(None, 0, 0, 0, None, 0, 0, 0, None, 0, 0, 0, 4, 4, 4, None, None, 0, 0.0, 0, 1, None, 0, None)
    assert rc == -1 and b"null operand" in lib.gdm_last_error()
    rc = lib.gdm_bce_with_logits(None, 0.0, 4, 1.0, None, None, 0, 0, None)
    assert rc == -1


def test_ops_refuse_cpu_tensors():
    import torch
    from gan_des_midi_music_gen_amd import ops
    with pytest.raises(ops.GdmError):
        ops.gemm(torch.zeros(2, 2), torch.zeros(2, 2))
    with pytest.raises(ops.GdmError):
        ops.adam_step(torch.zeros(4), torch.zeros(4), torch.zeros(4), torch.zeros(4), 1, 1e-3, 0.9, 0.999, 1e-8)


def test_build_stamp_covers_flags_and_compiler(monkeypatch):
    """build.py rebuilds every object when the flag set (or the compiler) differs from the one recorded beside the
    objects, and an experiment flag set marks the library (gdm_build_flavor)."""
    import importlib
    from gan_des_midi_music_gen_amd import build
    shipped = build._flags_stamp(build._hipcc())
    assert "--offload-arch=gfx950" in shipped and "GDM_EXPERIMENT_BUILD" not in shipped
    assert open(build.STAMP).read() == shipped, "the in-tree objects were built with the shipped flags"
    monkeypatch.setenv("GDM_HIPCC_FLAGS", "-DGDM_STAMPS")
    monkeypatch.setenv("GDM_BUILD_TAG", "stamp_test")
    variant = importlib.reload(build)
    try:
        assert "-DGDM_STAMPS" in variant._flags_stamp(variant._hipcc()) and "-DGDM_EXPERIMENT_BUILD=1" in variant.FLAGS
        assert variant.LIB.endswith("libgdm_hip_stamp_test.so") and variant.OBJ.endswith("_obj_stamp_test")
        assert variant._flags_stamp(variant._hipcc()) != shipped
    finally:
        monkeypatch.delenv("GDM_HIPCC_FLAGS")
        monkeypatch.delenv("GDM_BUILD_TAG")
        importlib.reload(build)
    from gan_des_midi_music_gen_amd import _lib
    assert _lib.load().gdm_build_flavor() == 0
