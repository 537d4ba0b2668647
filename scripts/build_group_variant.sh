#!/bin/bash
# Build an A/B variant of libtpc_mpc.so into ab/NAME/: the regular objects, except the GROUP units of the given
# horizons, recompiled with extra flags.   scripts/build_group_variant.sh NAME H "FLAGS" [H "FLAGS" ...]
# e.g.  scripts/build_group_variant.sh occ1 20 "-DTPC_GROUP_OCC=1" 40 "-DTPC_GROUP_OCC=1"   (repo root, after `make`)
set -e
NAME=$1; shift
CS=trajectory_controller_amd/csrc
LIB=trajectory_controller_amd/lib
mkdir -p ab/$NAME/obj
cp $LIB/obj/*.o ab/$NAME/obj/
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-slp-vectorize -Wall -Wno-unused-function"
while [ $# -ge 2 ]; do
  H=$1; EXTRA=$2; shift 2
  /opt/rocm/bin/hipcc $FLAGS $EXTRA -DTPC_GROUP_H=$H -c $CS/mpc_group_inst.hip -o ab/$NAME/obj/group_h$H.o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ab/$NAME/libtpc_mpc.so ab/$NAME/obj/*.o -ldl
echo "built ab/$NAME/libtpc_mpc.so"
