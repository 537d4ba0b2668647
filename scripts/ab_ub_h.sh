#!/bin/bash
# scripts/ab_ub_h.sh ROUNDS dtype H,H,... lib_dir...   : interleaved LANE_FMA kernel timing of several builds, several horizons
R=$1; D=$2; H=$3; shift 3
for i in $(seq $R); do
  for L in "$@"; do
    TPC_MPC_LIB=$PWD/$L/libtpc_mpc.so python scripts/lane_h.py $D $H 262144 lane_fma 2>&1 | grep -v amdgpu
  done
done
