#!/bin/bash
# scripts/ab_ub.sh ROUNDS dtype H lib_dir...   : interleaved LANE_FMA kernel timing of several builds on one box
R=$1; D=$2; H=$3; shift 3
for i in $(seq $R); do
  for L in "$@"; do
    TPC_MPC_LIB=$PWD/$L/libtpc_mpc.so python scripts/lane_h.py $D $H 262144 lane_fma 2>&1 | grep -v amdgpu
  done
done
