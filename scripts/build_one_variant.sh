#!/bin/bash
# Build an A/B variant of libtpc_mpc.so into ab/NAME/ with the resident single-solve unit (tpc_mpc_one.hip) recompiled
# with extra flags:   scripts/build_one_variant.sh NAME "FLAGS"     e.g. timing "-DTPC_ONE_TIMING"
set -e
NAME=$1; EXTRA=$2
CS=trajectory_controller_amd/csrc
LIB=trajectory_controller_amd/lib
mkdir -p ab/$NAME/obj
cp $LIB/obj/*.o ab/$NAME/obj/
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-slp-vectorize -Wall -Wno-unused-function"
/opt/rocm/bin/hipcc $FLAGS $EXTRA -c $CS/tpc_mpc_one.hip -o ab/$NAME/obj/one.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ab/$NAME/libtpc_mpc.so ab/$NAME/obj/*.o -ldl
echo "built ab/$NAME/libtpc_mpc.so"
