"""Builds libovplane_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU)."""
from __future__ import annotations

import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
SOURCES = ["ovp_api_ctx.hip", "ovp_api_point.hip", "ovp_api_rccl.hip", "ovp_api_plane.hip", "ovp_api_slam.hip", "k_feat.hip", "k_gram.hip", "k_ekf.hip", "k_init.hip", "k_tile.hip", "k_chol2.hip", "k_plane.hip", "k_plane2.hip", "k_triang.hip", "k_planefit.hip", "k_slam.hip", "k_dinit.hip"]
HEADERS = ["ovp_ctx.h", "ovp_dev.h", "ovp_kernels.h", "ovp_feat_model.h", "k_chol2.h", "k_plane2.h", "k_tile_body.h", "k_dpp.h", "k_slam.h", "k_dinit.h", os.path.join("..", "..", "include", "ovplane_hip.h")]
OUT = os.path.join(_HERE, "libovplane_hip.so")


def _stale():
    if not os.path.exists(OUT):
        return True
    t = os.path.getmtime(OUT)
    return any(os.path.getmtime(os.path.join(CSRC, f)) > t for f in SOURCES + HEADERS)


OBJ_DIR = os.path.join(CSRC, "_obj")


def build_lib(force=False, verbose=False):
    """One object per translation unit (compiled in parallel, rebuilt only when the source or a header is newer), one link."""
    if not force and not _stale():
        return OUT
    from concurrent.futures import ThreadPoolExecutor

    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    flags = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-value",
             "-I" + os.path.join(_HERE, "..", "include")] + os.environ.get("OVP_EXTRA_HIPCC_FLAGS", "").split()
    os.makedirs(OBJ_DIR, exist_ok=True)
    t_hdr = max(os.path.getmtime(os.path.join(CSRC, h)) for h in HEADERS)
    stamp = os.path.join(OBJ_DIR, "flags.txt")
    flag_key = " ".join(flags).replace(_HERE, ".")  # checkout-independent
    same_flags = os.path.exists(stamp) and open(stamp).read() == flag_key

    def one(src):
        obj = os.path.join(OBJ_DIR, src.replace(".hip", ".o"))
        path = os.path.join(CSRC, src)
        if force or not same_flags or not os.path.exists(obj) or os.path.getmtime(obj) < max(os.path.getmtime(path), t_hdr):
            cmd = [hipcc] + flags + ["-c", path, "-o", obj]
            if verbose:
                print(" ".join(cmd))
            subprocess.check_call(cmd)
        return obj

    with ThreadPoolExecutor(max_workers=min(len(SOURCES), os.cpu_count() or 4)) as ex:
        objs = list(ex.map(one, SOURCES))
    with open(stamp, "w") as f:
        f.write(flag_key)
    cmd = [hipcc, "--offload-arch=gfx950", "-fPIC", "-shared"] + os.environ.get("OVP_EXTRA_HIPCC_FLAGS", "").split() + objs + ["-o", OUT]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return OUT


HOST_DIR = os.path.join(CSRC, "host")
HOST_SOURCES = ["ov_plane_host.cpp", "ov_plane_updaters.cpp", "ov_plane_propagator.cpp", "ov_plane_zupt.cpp", "ov_plane_planefit.cpp", "ov_plane_io.cpp", "ov_plane_session.cpp", "host_capi.cpp"]
HOST_HEADERS = ["ov_plane_host.h", "ov_types.h"]
HOST_OUT = os.path.join(_HERE, "libovplane_host.so")


def build_host(force=False, verbose=False):
    """C++ host mirror of ov_plane's State / StateHelper / UpdaterMSCKF on top of the C-ABI (g++, links libovplane_hip.so)."""
    build_lib(force=False, verbose=verbose)
    stale = force or not os.path.exists(HOST_OUT)
    if not stale:
        t = os.path.getmtime(HOST_OUT)
        stale = any(os.path.getmtime(os.path.join(HOST_DIR, f)) > t for f in HOST_SOURCES + HOST_HEADERS) or os.path.getmtime(OUT) > t
    if not stale:
        return HOST_OUT
    cmd = ["g++", "-std=c++17", "-O3", "-fPIC", "-shared", "-I" + HOST_DIR, "-I" + os.path.join(_HERE, "..", "include")] + \
          [os.path.join(HOST_DIR, s) for s in HOST_SOURCES] + ["-o", HOST_OUT, "-L" + _HERE, "-lovplane_hip", "-Wl,-rpath,$ORIGIN"]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return HOST_OUT


def source_tree_hash():
    """sha256 over the sources libovplane_hip.so is built from (every .hip and header under csrc/, the C-ABI header), by name and
    content: the identity of the kernels a profile was taken on.  Profiles under profiles/ record it (tools/pmc_summary.py,
    tools/rocpd_stats.py) and bench.py quotes a committed counter file only when its hash is the running tree's - .git does not
    travel to the GPU box, so this is computed from the files themselves."""
    import hashlib

    h = hashlib.sha256()
    names = sorted(f for f in os.listdir(CSRC) if f.endswith((".hip", ".h")))
    paths = [os.path.join(CSRC, f) for f in names] + [os.path.join(_HERE, "..", "include", "ovplane_hip.h")]
    for path in paths:
        h.update(os.path.basename(path).encode() + b"\0")
        with open(path, "rb") as fh:
            h.update(fh.read())
        h.update(b"\0")
    return h.hexdigest()[:16]
