"""Per-frame binary trace of the update step.

One file = magic + any number of frame records; a record holds everything `ovp_msckf_update` reads (pose tables, covariance,
feature batch, options) and optionally what it returned (dx, accept mask, chi2, covariance after the update).  A frame dumped
next to the reference (where ROS + open_vins exist) can be replayed here and compared offline; the C++ twin is
ov_plane_amd/csrc/host/ov_plane_io.{h,cpp} (write_frame_trace / read_frame_trace).  All little-endian, matrices column-major.

  magic   8 bytes  "OVPTRC01"
  record  f64 timestamp | i32 C F M N
          f64 clone_q[C*4] clone_p[C*3] clone_q_fej[C*4] clone_p_fej[C*3] | i32 clone_id[C]
          f64 calib_q[4] calib_p[3] intrinsics[8] | i32 calib_id intr_id
          f64 P[N*N]
          f32 uv[F*M*2] | i32 clone_idx[F*M] n_meas[F] | f64 p_FinG[F*3]
          f64 sigma_px chi2_mult sigma_c | i32 do_fej do_calib_pose do_calib_intr has_outputs
          if has_outputs: f64 dx[N] | u8 accepted[F] | f64 chi2[F] | f64 P_after[N*N]
"""
from __future__ import annotations

import struct

import numpy as np

MAGIC = b"OVPTRC01"


def frame_from_scene(sc, outputs=None, timestamp=0.0):
    """Scene (ov_plane_amd.synth) -> dict in trace layout; outputs = dict(dx, accepted, chi2, P) or None."""
    f = dict(timestamp=float(timestamp), C=int(sc.C), F=int(sc.F), M=int(sc.uv.shape[1]), N=int(sc.N),
             clone_q=np.asarray(sc.clone_q, dtype=np.float64), clone_p=np.asarray(sc.clone_p, dtype=np.float64),
             clone_q_fej=np.asarray(sc.clone_q_fej, dtype=np.float64), clone_p_fej=np.asarray(sc.clone_p_fej, dtype=np.float64),
             clone_id=np.asarray(sc.ids["clones"], dtype=np.int32), calib_q=np.asarray(sc.calib_q, dtype=np.float64),
             calib_p=np.asarray(sc.calib_p, dtype=np.float64), intrinsics=np.asarray(sc.intr, dtype=np.float64),
             calib_id=int(sc.ids["calib"]) if sc.opts["do_calib_pose"] else -1,
             intr_id=int(sc.ids["intr"]) if sc.opts["do_calib_intr"] else -1, P=np.asarray(sc.P, dtype=np.float64),
             uv=np.asarray(sc.uv, dtype=np.float32), clone_idx=np.asarray(sc.clone_idx, dtype=np.int32),
             n_meas=np.asarray(sc.n_meas, dtype=np.int32), p_FinG=np.asarray(sc.p_FinG, dtype=np.float64),
             sigma_px=float(sc.opts["sigma_px"]), chi2_mult=float(sc.opts["chi2_mult"]), sigma_c=float(sc.opts["sigma_c"]),
             do_fej=int(sc.opts["do_fej"]), do_calib_pose=int(sc.opts["do_calib_pose"]), do_calib_intr=int(sc.opts["do_calib_intr"]))
    if outputs is not None:
        f.update(dx=np.asarray(outputs["dx"], dtype=np.float64), accepted=np.asarray(outputs["accepted"], dtype=np.uint8),
                 chi2=np.asarray(outputs["chi2"], dtype=np.float64), P_after=np.asarray(outputs["P"], dtype=np.float64))
    return f


def write_frames(path, frames):
    with open(path, "wb") as fh:
        fh.write(MAGIC)
        for f in frames:
            fh.write(struct.pack("<d4i", f["timestamp"], f["C"], f["F"], f["M"], f["N"]))
            for key in ("clone_q", "clone_p", "clone_q_fej", "clone_p_fej"):
                fh.write(np.ascontiguousarray(f[key], dtype="<f8").tobytes())
            fh.write(np.ascontiguousarray(f["clone_id"], dtype="<i4").tobytes())
            for key in ("calib_q", "calib_p", "intrinsics"):
                fh.write(np.ascontiguousarray(f[key], dtype="<f8").tobytes())
            fh.write(struct.pack("<2i", f["calib_id"], f["intr_id"]))
            fh.write(np.asfortranarray(f["P"], dtype="<f8").tobytes(order="F"))
            fh.write(np.ascontiguousarray(f["uv"], dtype="<f4").tobytes())
            fh.write(np.ascontiguousarray(f["clone_idx"], dtype="<i4").tobytes())
            fh.write(np.ascontiguousarray(f["n_meas"], dtype="<i4").tobytes())
            fh.write(np.ascontiguousarray(f["p_FinG"], dtype="<f8").tobytes())
            has = "dx" in f
            fh.write(struct.pack("<3d4i", f["sigma_px"], f["chi2_mult"], f["sigma_c"], f["do_fej"], f["do_calib_pose"],
                                 f["do_calib_intr"], int(has)))
            if has:
                fh.write(np.ascontiguousarray(f["dx"], dtype="<f8").tobytes())
                fh.write(np.ascontiguousarray(f["accepted"], dtype="u1").tobytes())
                fh.write(np.ascontiguousarray(f["chi2"], dtype="<f8").tobytes())
                fh.write(np.asfortranarray(f["P_after"], dtype="<f8").tobytes(order="F"))


def read_frames(path):
    frames = []
    with open(path, "rb") as fh:
        if fh.read(8) != MAGIC:
            raise ValueError("not an OVPTRC01 trace: %s" % path)

        def arr(dtype, n):
            a = np.frombuffer(fh.read(np.dtype(dtype).itemsize * n), dtype=dtype)
            if a.size != n:
                raise ValueError("truncated trace")
            return a.copy()

        while True:
            head = fh.read(8 + 16)
            if not head:
                break
            t, C, F, M, N = struct.unpack("<d4i", head)
            f = dict(timestamp=t, C=C, F=F, M=M, N=N)
            f["clone_q"] = arr("<f8", 4 * C).reshape(C, 4)
            f["clone_p"] = arr("<f8", 3 * C).reshape(C, 3)
            f["clone_q_fej"] = arr("<f8", 4 * C).reshape(C, 4)
            f["clone_p_fej"] = arr("<f8", 3 * C).reshape(C, 3)
            f["clone_id"] = arr("<i4", C)
            f["calib_q"], f["calib_p"], f["intrinsics"] = arr("<f8", 4), arr("<f8", 3), arr("<f8", 8)
            f["calib_id"], f["intr_id"] = struct.unpack("<2i", fh.read(8))
            f["P"] = arr("<f8", N * N).reshape(N, N, order="F")
            f["uv"] = arr("<f4", F * M * 2).reshape(F, M, 2)
            f["clone_idx"] = arr("<i4", F * M).reshape(F, M)
            f["n_meas"] = arr("<i4", F)
            f["p_FinG"] = arr("<f8", 3 * F).reshape(F, 3)
            (f["sigma_px"], f["chi2_mult"], f["sigma_c"], f["do_fej"], f["do_calib_pose"], f["do_calib_intr"],
             has) = struct.unpack("<3d4i", fh.read(24 + 16))
            if has:
                f["dx"] = arr("<f8", N)
                f["accepted"] = arr("u1", F)
                f["chi2"] = arr("<f8", F)
                f["P_after"] = arr("<f8", N * N).reshape(N, N, order="F")
            frames.append(f)
    return frames


def scene_from_frame(f):
    """Trace frame -> the Scene fields Context.state_upload / batch_upload_scene / opts_from_scene read."""
    from .synth import Scene

    ids = dict(clones=[int(i) for i in f["clone_id"]], calib=int(f["calib_id"]), intr=int(f["intr_id"]), N=int(f["N"]))
    return Scene(C=f["C"], F=f["F"], N=f["N"], ids=ids, clone_q=f["clone_q"], clone_p=f["clone_p"], clone_q_fej=f["clone_q_fej"],
                 clone_p_fej=f["clone_p_fej"], calib_q=f["calib_q"], calib_p=f["calib_p"], intr=f["intrinsics"], P=f["P"],
                 uv=f["uv"], clone_idx=f["clone_idx"], n_meas=f["n_meas"], p_FinG=f["p_FinG"],
                 opts=dict(sigma_px=f["sigma_px"], chi2_mult=f["chi2_mult"], sigma_c=f["sigma_c"], do_fej=bool(f["do_fej"]),
                           do_calib_pose=bool(f["do_calib_pose"]), do_calib_intr=bool(f["do_calib_intr"])))


def replay(path):
    """Replays every frame of a trace through the C-ABI on cuda:0 and compares with the recorded outputs (if any)."""
    from . import capi

    rows = []
    for k, f in enumerate(read_frames(path)):
        sc = scene_from_frame(f)
        ctx = capi.Context(sc.N, sc.C, max(sc.F, 1))
        ctx.cov_upload(sc.P)
        ctx.state_upload(sc)
        ctx.batch_upload_scene(sc, None)
        out = ctx.msckf_update(capi.opts_from_scene(sc))
        row = dict(frame=k, timestamp=f["timestamp"], n_feats=sc.F, accepted=int(out["accepted"].sum()))
        if "dx" in f:
            P = ctx.cov_download()
            d = np.sqrt(np.abs(np.diag(f["P_after"])))
            row.update(accept_mismatch=int((out["accepted"] != f["accepted"].astype(bool)).sum()),
                       max_abs_ddx=float(np.abs(out["dx"] - f["dx"]).max()),
                       max_rel_dP=float((np.abs(P - f["P_after"]) / np.outer(d, d)).max()))
        ctx.close()
        rows.append(row)
    return rows


if __name__ == "__main__":  # python -m ov_plane_amd.trace FILE
    import json
    import sys

    for r in replay(sys.argv[1]):
        print(json.dumps(r))
