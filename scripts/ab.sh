#!/bin/bash
# A/B two builds of libtpc_mpc.so on the SAME GPU box, interleaved: scripts/ab.sh ab/old ab/new [rounds]
A=$1; B=$2; R=${3:-3}
for i in $(seq $R); do
  for L in $A $B; do
    TPC_MPC_LIB=$PWD/$L/libtpc_mpc.so timeout -k 10 300 python bench.py --steps 5 --warmup 1 --no-cpu --no-fp32 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$L', round(d['value']/1e6,3), 'M/s  pg_ms', round(d['kernel_ms']['second'],3))"
  done
done
