#!/bin/bash
# tools/build/libmgcr_hip_timing.so: the library with csrc/gcr_resident.hip built -DMGCR_RES_TIMING (the kernel's own clock reads per
# phase; tools/collect_profiles.sh swaps it in for the resident-timing table).  Needs the regular build (objects in csrc/build).
set -e
cd "$(dirname "$0")/.."
C=mgpreconditionedgcr_amd/csrc
make -C $C -j8 > /dev/null
mkdir -p tools/build
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -Wall -Wno-unused-function -Wno-unused-value -Wno-unused-result \
    -DMGCR_RES_TIMING -c $C/gcr_resident.hip -o tools/build/gcr_resident_timing.o
objs=$(ls $C/build/*.o | grep -v gcr_resident.o)
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o tools/build/libmgcr_hip_timing.so $objs tools/build/gcr_resident_timing.o
ls -la tools/build/libmgcr_hip_timing.so
