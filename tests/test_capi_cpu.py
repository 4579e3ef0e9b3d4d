"""CPU tests of the boundary: the C-ABI library builds, loads, exports every declared symbol, and fails loudly
without a gfx950 device (no compute is attempted here)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    txt = open(os.path.join(ROOT, "include", "ovplane_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(ovp_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol(hiplib):
    L = hiplib.lib()
    names = _declared_symbols()
    assert len(names) >= 25
    for n in names:
        assert hasattr(L, n), "missing export %s" % n
    assert set(hiplib.EXPORTS) == set(names)


def test_version_and_error_strings(hiplib):
    L = hiplib.lib()
    assert b"gfx950" in L.ovp_version()
    assert L.ovp_error_string(0) == b"ok"
    assert L.ovp_error_string(-5) == b"no usable HIP device"


def test_chi2_quantile_host_function(hiplib):
    tab = np.load(os.path.join(ROOT, "tests", "golden", "chi2_095_table.npy"))
    got = np.array([hiplib.lib().ovp_chi2_quantile_095(k) for k in range(1, 1001)])
    assert np.abs(got - tab[1:]).max() / tab[1:].max() < 1e-12


def test_context_creation_fails_loudly_without_gpu(hiplib):
    import torch

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(hiplib.OvpError):
        hiplib.Context(64, 4, 8)


def test_product_package_does_not_import_oracle():
    """The oracle is test infrastructure; nothing under ov_plane_amd/ may reference it."""
    for dirpath, _, files in os.walk(os.path.join(ROOT, "ov_plane_amd")):
        for fn in files:
            if fn.endswith((".py", ".hip", ".h", ".cpp")):
                txt = open(os.path.join(dirpath, fn), errors="ignore").read()
                assert "pyoracle" not in txt and "ovp_oracle" not in txt and "np_ref" not in txt, fn


def test_struct_layouts_match_header(hiplib):
    assert C.sizeof(hiplib.UpdateOpts) == 40
    assert C.sizeof(hiplib.UpdateInfo) == 32
    assert C.sizeof(hiplib.FeatureBatch) == 40
    assert C.sizeof(hiplib.StateTables) == 8 + 5 * 8 + 7 * 8 + 8 + 8 * 8 + 8
