#!/bin/bash
# builds tools/ab/libovplane_hip_k1stamps.so = the product library with k_feat.hip compiled under -DOVP_K1_STAMPS (for tools/k1_stamps.py)
set -e
cd "$(dirname "$0")/.."
python -c "
import sys; sys.path.insert(0,'.')
from ov_plane_amd.build import build_lib; build_lib()"
mkdir -p tools/ab
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value -Iinclude -DOVP_K1_STAMPS -c ov_plane_amd/csrc/k_feat.hip -o /tmp/k_feat_stamps.o -Rpass-analysis=kernel-resource-usage 2>&1 | grep -E "VGPRs Spill|ScratchSize" | sort | uniq -c
/opt/rocm/bin/hipcc --offload-arch=gfx950 -fPIC -shared $(ls ov_plane_amd/csrc/_obj/*.o | grep -v "k_feat.o") /tmp/k_feat_stamps.o -o tools/ab/libovplane_hip_k1stamps.so
