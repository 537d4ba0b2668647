#!/bin/bash
# Every rocprofv3 summary DESIGN.md quotes, in one gpurun call (run from the repo root):
#   bash scripts/profile_all.sh r03        -> gpurun_out/r02_<config>_{kernel_stats,pmc}.csv  (copy into profiles/)
# Each config = scripts/profile.sh: one --kernel-trace --stats run, then separate --pmc runs.
# bench.py runs with --no-pipelined: the default run appends a two-batches-in-flight leg after its timed region,
# whose overlapping dispatches would be averaged into the same per-kernel rows.
R=${1:-r03}
P="bash scripts/profile.sh"
$P ${R}_headline  python3 bench.py --steps 10 --warmup 2 --no-cpu --no-fp32 --no-pipelined --no-bit-exact --no-config2 --no-config5
$P ${R}_bitexact  python3 bench.py --steps 10 --warmup 2 --algo lane --no-cpu --no-fp32 --no-pipelined --no-config2 --no-config5
$P ${R}_config2   python3 bench.py --steps 20 --warmup 2 --batch 4096 --horizon 10 --algo wave --no-cpu --no-fp32 --no-pipelined
$P ${R}_fp32      python3 bench.py --steps 10 --warmup 2 --dtype f32 --no-cpu --no-pipelined --no-config2 --no-config5
$P ${R}_h30       python3 scripts/lane_h.py f64 30 262144 lane_fma
$P ${R}_h40       python3 scripts/lane_h.py f64 40 262144 lane_fma
$P ${R}_wave2     python3 scripts/lane_h.py f64 40 8192 wave
$P ${R}_general   python3 scripts/general_rate.py 2 20 lane
$P ${R}_follow    python3 scripts/follow_rate.py 262144 10
ls gpurun_out/${R}_*_kernel_stats.csv
