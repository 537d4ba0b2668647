#!/bin/bash
# interleaved A/B of several builds on one box: scripts/probes/ab_libs.sh ROUNDS dir...   (headline batch, LANE_FMA, PG kernel time)
R=$1; shift
for i in $(seq $R); do
  for L in "$@"; do
    echo "== $L"
    TPC_MPC_LIB=$PWD/$L/libtpc_mpc.so timeout -k 10 120 python scripts/probes/asm_stats.py 2>&1 | tail -2
  done
done
