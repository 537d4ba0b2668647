#!/bin/bash
# quick GPU check of the LANE_FMA family + the bench line (used while iterating on a kernel)
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_ub_gpu.py -x -q -m gpu 2>&1 | tail -15 > gpurun_out/quick_tests.txt
rc=$?
cat gpurun_out/quick_tests.txt
[ $rc -eq 0 ] || exit $rc
timeout -k 10 400 python bench.py > gpurun_out/quick_bench.json 2> gpurun_out/quick_bench.err || { tail -20 gpurun_out/quick_bench.err; exit 1; }
python - <<'PY'
import json
d = json.loads(open("gpurun_out/quick_bench.json").read().strip().splitlines()[-1])
for k in ("value", "ms_per_step", "kernel_ms", "roofline", "alu", "max_abs_du_vs_dlib", "vs_bit_exact", "pipelined"):
    print(k, d.get(k))
PY
