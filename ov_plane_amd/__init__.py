"""ov_plane_amd - Python side of the MI355X-native MSCKF(+plane) update path for rpng/ov_plane.

The product is the C-ABI library `libovplane_hip.so` (`include/ovplane_hip.h`; kernels and entry points under `csrc/`) and the C++
mirror of ov_plane's updater surface above it (`csrc/host/`, `libovplane_host.so`).  The modules here are plumbing around them:

  build        hipcc / g++ recipes (in-tree, gfx950), `source_tree_hash()` - the identity of the kernel sources a profile was taken on
  capi         ctypes binding of the C-ABI (what the GPU tests and bench.py call); fails loudly when the library is not built
  hostlib      ctypes access to the C++ host mirror's test harness
  dist         host logic of the feature-sharded multi-GPU update over torch.distributed (the native form is ovp_msckf_update_sharded)
  synth, sim   seeded scene generator / restated simulator
  trace        per-frame binary trace format (inputs of one update + its outputs) and its replay
  closed_loop  the slice of VioManager a closed loop needs, over the host mirror

Nothing in this package imports the oracle (test infrastructure under `oracle/`); there is no CPU fallback of the device path."""

__version__ = "0.6"
