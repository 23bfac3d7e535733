#!/bin/bash
# experiment build of the A* kernel without touching the shipped objects: scripts/build_variant.sh <name> "<extra flags>"
# -> megagta_amd/libmegagta_hip_<name>.so (select with MEGAGTA_HIP_LIB); the other objects are taken as they are
set -e
name=$1; extra=$2
cd "$(dirname "$0")/../megagta_amd/csrc"
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -ffp-contract=off -Wall -Wno-unused-function -Wno-unused-result -Wno-pass-failed $extra -c astar.hip -o astar_$name.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC ctx.o sdbg_build.o graph.o findstart.o denovo.o ingest.o probe.o astar_$name.o -o ../libmegagta_hip_$name.so
echo "built libmegagta_hip_$name.so"
