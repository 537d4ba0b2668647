#!/bin/bash
# Every rocprofv3 summary DESIGN.md / profiles/README.md quote (run from the repo root, through gpurun):
#   bash scripts/profile_all.sh r05 [a|b|c|d|e|all]   -> gpurun_out/r05_<tag>_{kernel_stats,pmc}.csv  (copy into profiles/)
# Each tag = scripts/profile.sh: one --kernel-trace --stats run, then separate --pmc runs.  Three parts, because one
# gpurun call is limited to 20 minutes:  a = the bench.py workloads, b = long horizons / general form / follow,
# c = the GROUP family (round 4) and config 5, d = GROUP at 262 144 x N = 30 / 40 and for the general form, the bit-exact family G lanes per instance.
# bench.py runs with --no-pipelined: the default run appends a two-batches-in-flight leg after its timed region,
# whose overlapping dispatches would be averaged into the same per-kernel rows.
R=${1:-r05}
PART=${2:-all}
P="bash scripts/profile.sh"
LOG=gpurun_out/prof_${R}_${PART}.log
: > $LOG
if [ $PART = a ] || [ $PART = all ]; then
$P ${R}_headline  python3 bench.py --steps 10 --warmup 2 --no-cpu --no-fp32 --no-pipelined --no-bit-exact --no-config1 --no-config2 --no-config5 --no-config4 --no-mid >> $LOG 2>&1
$P ${R}_bitexact  python3 bench.py --steps 10 --warmup 2 --algo lane --no-cpu --no-fp32 --no-pipelined --no-config1 --no-config2 --no-config5 --no-config4 --no-mid >> $LOG 2>&1
$P ${R}_config2   python3 bench.py --steps 20 --warmup 2 --batch 4096 --horizon 10 --algo wave --no-cpu --no-fp32 --no-pipelined >> $LOG 2>&1
$P ${R}_fp32      python3 bench.py --steps 10 --warmup 2 --dtype f32 --no-cpu --no-pipelined --no-config1 --no-config2 --no-config5 --no-config4 --no-mid >> $LOG 2>&1
fi
if [ $PART = b ] || [ $PART = all ]; then
$P ${R}_scan       python3 scripts/lane_h.py f64 40 8192 wave >> $LOG 2>&1
$P ${R}_wave2      python3 scripts/general_rate.py 2 40 wave 8192 >> $LOG 2>&1
$P ${R}_general    python3 scripts/general_rate.py 2 20 lane >> $LOG 2>&1
$P ${R}_generalfma python3 scripts/general_rate.py 2 20 lane_fma >> $LOG 2>&1
$P ${R}_follow     python3 scripts/follow_rate.py 262144 10 >> $LOG 2>&1
fi
if [ $PART = c ] || [ $PART = all ]; then
$P ${R}_group_h20  python3 scripts/lane_h.py f64 20 16384 group >> $LOG 2>&1
$P ${R}_group_h20_64k python3 scripts/lane_h.py f64 20 65536 group >> $LOG 2>&1
$P ${R}_group_h40  python3 scripts/lane_h.py f64 40 16384 group >> $LOG 2>&1
$P ${R}_group_h10  python3 scripts/lane_h.py f64 10 32768 group >> $LOG 2>&1
$P ${R}_config5    python3 scripts/mixed_horizons.py f64fast >> $LOG 2>&1
$P ${R}_config5_shipped python3 scripts/mixed_horizons.py f64 >> $LOG 2>&1   # with AUTO's guarantee: the presolve beside the bins
fi
if [ $PART = d ] || [ $PART = all ]; then   # GROUP where AUTO takes it at the full batch (long horizons), and the general form
$P ${R}_group_h40_256k python3 scripts/lane_h.py f64 40 262144 group >> $LOG 2>&1
$P ${R}_group_h30_256k python3 scripts/lane_h.py f64 30 262144 group >> $LOG 2>&1
$P ${R}_groupg_h40     python3 scripts/general_rate.py 2 40 group 16384 >> $LOG 2>&1
$P ${R}_groupg_h20     python3 scripts/general_rate.py 2 20 group 16384 >> $LOG 2>&1
$P ${R}_lanex_h40      python3 scripts/lane_h.py f64 40 16384 lane >> $LOG 2>&1
$P ${R}_lanex_h20      python3 scripts/lane_h.py f64 20 16384 lane >> $LOG 2>&1
fi
if [ $PART = e ] || [ $PART = all ]; then   # fp32 GROUP, two wavefronts per SIMD (what AUTO runs for a 262 144-instance fp32 batch at N = 20 / 40)
$P ${R}_group_f32_h20_256k python3 scripts/lane_h.py f32 20 262144 group >> $LOG 2>&1
$P ${R}_group_f32_h40_256k python3 scripts/lane_h.py f32 40 262144 group >> $LOG 2>&1
$P ${R}_f32_2m             python3 scripts/lane_h.py f32 20 2097152 lane_fma >> $LOG 2>&1   # config 4's whole batch on one GPU: LANE_FMA fp32 at its steady state
fi
ls gpurun_out/${R}_*_kernel_stats.csv
python3 -c "import hashlib;print(hashlib.sha256(open('trajectory_controller_amd/lib/libtpc_mpc.so','rb').read()).hexdigest())"
