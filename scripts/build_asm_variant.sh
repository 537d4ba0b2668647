#!/bin/bash
# Build a variant of libtpc_mpc.so into ab/NAME/ whose hand-written headline kernel is regenerated with other arguments:
#   scripts/build_asm_variant.sh NAME "GEN ARGS"     e.g.  scripts/build_asm_variant.sh diag4 "4 4"   (from the repo root, after `make`)
set -e
NAME=$1; ARGS=$2
CS=trajectory_controller_amd/csrc
LIB=trajectory_controller_amd/lib
mkdir -p ab/$NAME/obj ab/$NAME/inc
cp $LIB/obj/*.o ab/$NAME/obj/
# (a sixth generator argument of 10 regenerates the N = 10 kernel's header instead of the headline kernel's)
set -- $ARGS
if [ "${6:-20}" = "10" ]; then
  python3 scripts/gen_ub_pg_asm.py $ARGS > ab/$NAME/inc/mpc_ub_pg_asm_h10.h; cp $CS/mpc_ub_pg_asm.h ab/$NAME/inc/
else
  python3 scripts/gen_ub_pg_asm.py $ARGS > ab/$NAME/inc/mpc_ub_pg_asm.h; cp $CS/mpc_ub_pg_asm_h10.h ab/$NAME/inc/
fi
cp $CS/mpc_ub_asm.h $CS/mpc_ub_asm_inst.hip ab/$NAME/inc/
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-slp-vectorize -Wall -Wno-unused-function"
/opt/rocm/bin/hipcc $FLAGS -I$CS -Iinclude -c ab/$NAME/inc/mpc_ub_asm_inst.hip -o ab/$NAME/obj/ub_asm.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ab/$NAME/libtpc_mpc.so ab/$NAME/obj/*.o -ldl
echo "built ab/$NAME/libtpc_mpc.so"
