#!/bin/bash
# scripts/ab_multi.sh ROUNDS lib_dir...   : interleaved bench of several builds on one box (fp64 + fp32 legs)
R=$1; shift
for i in $(seq $R); do
  for L in "$@"; do
    TPC_MPC_LIB=$PWD/$L/libtpc_mpc.so timeout -k 10 300 python bench.py --steps 5 --warmup 1 --no-cpu | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$L', round(d['value']/1e6,3), 'M/s  pg_ms', round(d['kernel_ms']['second'],3), ' cd_ms', round(d['kernel_ms']['first'],3), ' fp32', round(d['fp32']['value']/1e6,2))"
  done
done
