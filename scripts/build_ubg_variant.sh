#!/bin/bash
# Build an A/B variant of libtpc_mpc.so into ab/NAME/ with the general-form LANE_FMA units of the given horizons
# recompiled with extra flags:   scripts/build_ubg_variant.sh NAME H "FLAGS" [H "FLAGS" ...]
set -e
NAME=$1; shift
CS=trajectory_controller_amd/csrc
LIB=trajectory_controller_amd/lib
mkdir -p ab/$NAME/obj
cp $LIB/obj/*.o ab/$NAME/obj/
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-slp-vectorize -Wall -Wno-unused-function"
while [ $# -ge 2 ]; do
  H=$1; EXTRA=$2; shift 2
  /opt/rocm/bin/hipcc $FLAGS $EXTRA -DTPC_UBG_H=$H -c $CS/mpc_ubg_inst.hip -o ab/$NAME/obj/ubg_h$H.o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ab/$NAME/libtpc_mpc.so ab/$NAME/obj/*.o -ldl
echo "built ab/$NAME/libtpc_mpc.so"
