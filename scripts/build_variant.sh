#!/bin/bash
# Build an A/B variant of libtpc_mpc.so into ab/NAME/: the regular objects, except the LANE units of the
# given horizons, which are recompiled with extra flags.   scripts/build_variant.sh NAME H "FLAGS" [H "FLAGS" ...]
# e.g.  scripts/build_variant.sh kv12 40 "-DTPC_KV_STEPS=12"     (run from the repo root, after `make`)
set -e
NAME=$1; shift
CS=trajectory_controller_amd/csrc
LIB=trajectory_controller_amd/lib
mkdir -p ab/$NAME/obj
cp $LIB/obj/*.o ab/$NAME/obj/
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-slp-vectorize -Wall -Wno-unused-function"
while [ $# -ge 2 ]; do
  H=$1; EXTRA=$2; shift 2
  /opt/rocm/bin/hipcc $FLAGS $EXTRA -DTPC_SCHED_NAME='"default"' -DTPC_LANE_H=$H -c $CS/mpc_lane_inst.hip -o ab/$NAME/obj/lane_h$H.o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ab/$NAME/libtpc_mpc.so ab/$NAME/obj/*.o -ldl
echo "built ab/$NAME/libtpc_mpc.so"
