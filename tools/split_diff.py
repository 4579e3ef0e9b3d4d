"""k_chol2's split solve against the one-workgroup solve on the frame of test_plane_solve_on_two_workgroups...: the differences themselves"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ov_plane_amd import capi
from ov_plane_amd.synth import make_scene
from oracle import pyoracle
pyoracle.build()
sc = make_scene(C=30, F=360, seed=21, n_planes=6, feats_per_plane=40, planes_in_state_frac=0.5, chi2_mult=1.0)
ref = pyoracle.msckf_plane_update(sc)
outs = []
for h in ("0", "5", "8"):
    os.environ["OVP_C2_SPLIT"] = h
    ctx = capi.Context(sc.N, sc.C, sc.F)
    ctx.cov_upload(sc.P); ctx.state_upload(sc); ctx.batch_upload_scene(sc)
    out = ctx.plane_update(capi.opts_from_scene(sc), sc.plane_id, sc.cp, sc.cp_fej, sc.plane_state_id, force_decision=ref["plane_ok"].astype(np.uint8))
    out["P"] = ctx.cov_download(); outs.append(out); ctx.close()
b = outs[0]
print("lib", os.environ.get("OVP_LIB_AB", "product"))
print(" single vs oracle: dx %.3e chi2 %s" % (np.abs(b["dx"] - ref["plane_dx"]).max() if "plane_dx" in ref else -1, np.abs(b["chi2"] - ref["plane_chi2"])))
for h, o in zip(("5", "8"), outs[1:]):
    print(" split %s vs single: chi2 diff %s | dx diff per plane %s" % (h, np.abs(o["chi2"] - b["chi2"]), np.abs(o["dx"] - b["dx"]).reshape(len(o["chi2"]), -1).max(axis=1)))
