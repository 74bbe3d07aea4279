#!/bin/bash
# experiment build of the library with per-workgroup timing stamps in k_tops_round
# (tools/exp_tops_timing.py); output tools/bin/libgklt_timing$GKLT_SUFFIX.so, never loaded by the product
set -e
cd "$(dirname "$0")/.."
mkdir -p tools/bin/obj
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -fno-fast-math -DGKLT_TOPS_TIMING $GKLT_EXTRA"
/opt/rocm/bin/hipcc $F -x hip -c gslam_amd/csrc/gklt_select.hip -o tools/bin/obj/gklt_select_timing.o
OBJS=$(ls gslam_amd/build/*.o | grep -v gklt_select)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OBJS tools/bin/obj/gklt_select_timing.o -o tools/bin/libgklt_timing$GKLT_SUFFIX.so
