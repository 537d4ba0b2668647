#!/bin/bash
# Builds and runs examples/solve_one_latency.c against the in-tree library (on the GPU box): scripts/latency.sh [reps]
set -e
cd "$(dirname "$0")/.."
L=trajectory_controller_amd/lib
mkdir -p gpurun_out
gcc -std=c99 -O2 -Iinclude examples/solve_one_latency.c -o gpurun_out/solve_one_latency -L$L -ltpc_mpc -Wl,-rpath,$PWD/$L
[ -n "$TPC_MPC_LIB_DIR" ] && export LD_LIBRARY_PATH=$TPC_MPC_LIB_DIR
./gpurun_out/solve_one_latency ${1:-3000} $2
