#!/bin/bash
# builds tools/ab/libovplane_hip_c2stamps.so = the product library with k_chol2.hip compiled under -DOVP_C2_STAMPS: the cycle stamps of
# the plane loop's factorization kernel (OVP_PL_STAMPS=1 tail only, 2 per step, 3 + the publication phase), e.g.
#   OVP_LIB_AB=tools/ab/libovplane_hip_c2stamps.so OVP_PL_STAMPS=2 python bench.py --workload config3 --steps 2 --warmup 1 --no-cpu-baseline --no-extras
set -e
cd "$(dirname "$0")/.."
python -c "
import sys; sys.path.insert(0,'.')
from ov_plane_amd.build import build_lib; build_lib()"
mkdir -p tools/ab
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value -Iinclude -DOVP_C2_STAMPS -c ov_plane_amd/csrc/k_chol2.hip -o /tmp/k_chol2_stamps.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -fPIC -shared $(ls ov_plane_amd/csrc/_obj/*.o | grep -v "k_chol2.o") /tmp/k_chol2_stamps.o -o tools/ab/libovplane_hip_c2stamps.so
