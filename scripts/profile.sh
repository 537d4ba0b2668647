#!/bin/bash
# Profiles of one command on the GPU box (run through gpurun from the repo root):
#   bash scripts/profile.sh TAG python3 bench.py --steps 2 --warmup 1 --no-cpu --no-fp32
# Writes gpurun_out/prof_TAG_{trace,fetch,write,sq}/ and the two judged summaries
# gpurun_out/TAG_kernel_stats.csv / gpurun_out/TAG_pmc.csv (copy them into profiles/).
# The PMC passes are separate runs with --kernel-trace only, as gpurun requires.
set -e
TAG=$1; shift
ROOT=$PWD
export TMPDIR=/tmp
cd /tmp
OUT=$ROOT/gpurun_out
mkdir -p $OUT
run() { # name, rocprof args...
  local name=$1; shift
  (cd $ROOT && timeout -k 10 400 rocprofv3 "$@" --output-format csv -d $OUT/prof_${TAG}_$name -- "${CMD[@]}") > $OUT/prof_${TAG}_$name.log 2>&1
}
CMD=("$@")
run trace --kernel-trace --stats
run fetch --kernel-trace --pmc FETCH_SIZE
run write --kernel-trace --pmc WRITE_SIZE
run sq --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS
[ -n "$TPC_PROFILE_SQ2" ] && run sq2 --kernel-trace --pmc SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE || true
cd $ROOT
python3 scripts/pmc_summary.py $TAG "${CMD[*]}" > $OUT/${TAG}_summary.txt
