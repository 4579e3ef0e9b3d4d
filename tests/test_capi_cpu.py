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


def test_host_givens_operations_match_restatement():
    """UpdaterHelper / UpdaterPlane ::nullspace_project_inplace and ::measurement_compress_inplace of the C++ host mirror
    (update/UpdaterHelper.cpp:515-579, update/UpdaterPlane.cpp:483-552) are pure host arithmetic: checked here without a GPU
    against the numpy restatement of the same rotation sequence."""
    import numpy as np

    from ov_plane_amd.build import build_host
    from oracle import np_ref

    build_host()
    from ov_plane_amd import hostlib

    rng = np.random.default_rng(12)
    rows, cols = 33, 14
    H_f, H_x, H_cp, res = (rng.standard_normal((rows, 3)), rng.standard_normal((rows, cols)), rng.standard_normal((rows, 3)),
                           rng.standard_normal(rows))
    # nullspace, with and without the plane Jacobian
    Hx, Hcp, r = hostlib.run_plane_givens(0, H_f, H_x, H_cp, res)
    eHx, er, eHcp = np_ref.nullspace_project_inplace(H_f, H_x, res, H_cp)
    assert Hx.shape == (rows - 3, cols)
    assert max(np.abs(Hx - eHx).max(), np.abs(Hcp - eHcp).max(), np.abs(r - er).max()) < 1e-13
    Hx2, _, r2 = hostlib.run_plane_givens(0, H_f, H_x, None, res)
    assert max(np.abs(Hx2 - eHx).max(), np.abs(r2 - er).max()) < 1e-13
    # compression
    Hx, Hcp, r = hostlib.run_plane_givens(1, None, H_x, H_cp, res)
    eHx, er, eHcp = np_ref.measurement_compress_inplace(H_x, res, H_cp)
    assert Hx.shape == (cols, cols)
    assert max(np.abs(Hx - eHx).max(), np.abs(Hcp - eHcp).max(), np.abs(r - er).max()) < 1e-13
    assert np.abs(Hx.T @ Hx - H_x.T @ H_x).max() < 1e-12
    # rows <= cols: untouched (UpdaterHelper.cpp:551-552)
    Hx, _, r = hostlib.run_plane_givens(1, None, H_x[:10], None, res[:10])
    assert Hx.shape == (10, cols) and np.abs(Hx - H_x[:10]).max() == 0.0 and np.abs(r - res[:10]).max() == 0.0


def test_no_dpp_read_after_valu_write_hazard_in_the_built_kernels():
    """The broadcast-in-FMA chains of k_chol2 / K1 are inline asm: the compiler's hazard recognizer does not see them, so the two
    wait states a DPP read needs behind a VALU write of its operand are checked in the final machine code of every build
    (tools/check_dpp_hazard.py)."""
    import importlib.util
    import glob

    from ov_plane_amd.build import OBJ_DIR, build_lib

    spec = importlib.util.spec_from_file_location("check_dpp_hazard", os.path.join(ROOT, "tools", "check_dpp_hazard.py"))
    chk = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(chk)
    # the checker itself: a producer one wait state in front of the DPP read is found, two (s_nop 1) are fine, another register is fine
    fn = ["0000000000001000 <k>:"]
    bad, n = chk.check_disassembly(fn + ["\tv_fma_f64 v[4:5], v[0:1], v[2:3], v[0:1]// 0", "\ts_nop 0// 0",
                                        "\tv_fmac_f64_dpp v[10:11], v[4:5], v[6:7] row_newbcast:3 row_mask:0xf bank_mask:0xf// 0"], "t")
    assert n == 1 and len(bad) == 1
    bad, n = chk.check_disassembly(fn + ["\tv_fma_f64 v[4:5], v[0:1], v[2:3], v[0:1]// 0", "\ts_nop 1// 0",
                                        "\tv_fmac_f64_dpp v[10:11], v[4:5], v[6:7] row_newbcast:3 row_mask:0xf bank_mask:0xf// 0"], "t")
    assert n == 1 and not bad
    bad, n = chk.check_disassembly(fn + ["\tv_mov_b32_e32 v5, v9// 0", "\tv_mov_b32_dpp v1, v5 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf// 0"], "t")
    assert len(bad) == 1
    bad, n = chk.check_disassembly(fn + ["\tv_mov_b32_e32 v6, v9// 0", "\tv_mov_b32_dpp v1, v5 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf// 0"], "t")
    assert not bad
    build_lib()
    from ov_plane_amd.build import SOURCES

    want = sorted(os.path.join(OBJ_DIR, s.replace(".hip", ".o")) for s in SOURCES)
    if not all(os.path.exists(o) for o in want):  # a library that is up to date beside a cleaned object directory
        build_lib(force=True)
    objs = sorted(glob.glob(os.path.join(OBJ_DIR, "*.o")))
    assert objs == want, (objs, want)   # exactly the translation units the shipped library is linked from
    n_total = 0
    for o in objs:
        bad, n, ncos = chk.check_object(o)
        assert not bad, bad
        # every kernel translation unit carries a gfx950 code object; the entry-point files (ovp_api_*.hip) may be host code only
        assert ncos >= 1 or os.path.basename(o).startswith("ovp_api_"), o
        n_total += n
    assert n_total > 10000  # the chains are there (k_chol2 alone holds ~16 K DPP instructions)
