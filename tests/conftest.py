import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _library_flavor():
    """0 = the shipped build; != 0 = libgdm_hip.so was built with experiment switches (GDM_HIPCC_FLAGS: instrumented or
    deliberately-wrong variants).  None when the library cannot be loaded (the gpu tests then fail loudly themselves)."""
    try:
        from gan_des_midi_music_gen_amd import _lib
        return int(_lib.load().gdm_build_flavor())
    except Exception:
        return None


def pytest_collection_modifyitems(config, items):
    """Parity results from an experiment build mean nothing: skip the gpu tests, with the reason, instead of passing or
    failing on a library that is not the product."""
    if not any("gpu" in it.keywords for it in items):
        return
    flavor = _library_flavor()
    if flavor in (0, None) or os.environ.get("GDM_TEST_ALLOW_EXPERIMENT") == "1":
        return
    skip = pytest.mark.skip(reason=f"libgdm_hip.so is an experiment build (gdm_build_flavor() = {flavor}, "
                                   "GDM_HIPCC_FLAGS): rebuild with `python -m gan_des_midi_music_gen_amd.build`")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")
