#!/bin/bash
# scripts/ab_stats.sh ROUNDS lib_dir... : interleaved fp64 bench of several builds on one box, with lane statistics
R=$1; shift
for i in $(seq $R); do
  for L in "$@"; do
    TPC_MPC_LIB=$PWD/$L/libtpc_mpc.so timeout -k 10 300 python bench.py --steps 5 --warmup 1 --no-cpu --no-fp32 --inflight 1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); s=d['lane_stats']; print('$L', round(d['value']/1e6,3), 'M/s  pg_ms', round(d['kernel_ms']['second'],3), ' wave_iters', s['wave_iterations'], ' refills', s['refill_blocks'], ' util', round(s['lane_utilisation'],4))"
  done
done
