#!/bin/bash
# A/B on one box: the hand-written kernel against the compiler's (ab/noasm, scripts/build_ub_variant.sh noasm 20 "-DTPC_UB_NO_ASM")
for i in 1 2 3; do
  for L in trajectory_controller_amd/lib ab/noasm; do
    echo "== $L"
    TPC_MPC_LIB=$PWD/$L/libtpc_mpc.so timeout -k 10 120 python scripts/loopcost.py 20 f64 65536 lane_fma 2>&1 | grep variant
    TPC_MPC_LIB=$PWD/$L/libtpc_mpc.so timeout -k 10 120 python scripts/probes/asm_stats.py 2>&1 | tail -1
  done
done
