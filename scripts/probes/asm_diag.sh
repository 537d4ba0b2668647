#!/bin/bash
# Where the hand-written headline kernel's time goes: diagnostic builds (scripts/build_asm_variant.sh diagK "4 K") report one quantity
# per wavefront in place of the wave-iteration statistic -- shader cycles of: 1 ticket wait, 2 queue-entry wait, 3 record wait, 4 refill
# passes, 5 the iteration loop, 6 the whole kernel body; 7: the body in 100 MHz ticks (6 / 7 x 100 MHz = the clock the kernel ran at)
for k in "$@"; do
  echo "== diag $k"
  TPC_MPC_LIB=$PWD/ab/diag$k/libtpc_mpc.so timeout -k 10 120 python scripts/probes/asm_stats.py 2>&1 | tail -2
done
echo "== shipped"
timeout -k 10 120 python scripts/probes/asm_stats.py 2>&1 | tail -2
