#!/bin/bash
# What the driver runs at round end, in its order, on one GPU box (through gpurun from the repo root):
#   bash scripts/driver_sequence.sh > gpurun_out/driver_sequence.txt 2>&1     (copy to profiles/rNN_driver_sequence.txt)
# The bench line is also left by itself in gpurun_out/bench_default.json.
set -o pipefail
mkdir -p gpurun_out
echo "# library sha256 $(sha256sum trajectory_controller_amd/lib/libtpc_mpc.so | cut -d' ' -f1)"
echo "== python -m pytest tests -x -q -m gpu"
python -m pytest tests -x -q -m gpu 2>&1 | tail -4 || exit 1
echo "== __graft_entry__.smoke()"
python -c "import __graft_entry__ as g; g.smoke()" || exit 1
echo "== python bench.py --gpus 1 --steps 20 --warmup 5"
python bench.py --gpus 1 --steps 20 --warmup 5 2> gpurun_out/bench_default.err | tee gpurun_out/bench_default.json
