P="bash scripts/profile.sh"
R=r03
$P ${R}_h30       python3 scripts/lane_h.py f64 30 262144 lane_fma > gpurun_out/prof_b.log 2>&1
$P ${R}_h40       python3 scripts/lane_h.py f64 40 262144 lane_fma >> gpurun_out/prof_b.log 2>&1
$P ${R}_scan      python3 scripts/lane_h.py f64 40 8192 wave >> gpurun_out/prof_b.log 2>&1
$P ${R}_wave2     python3 scripts/general_rate.py 2 40 wave 8192 >> gpurun_out/prof_b.log 2>&1
$P ${R}_general   python3 scripts/general_rate.py 2 20 lane >> gpurun_out/prof_b.log 2>&1
$P ${R}_generalfma python3 scripts/general_rate.py 2 20 lane_fma >> gpurun_out/prof_b.log 2>&1
$P ${R}_follow    python3 scripts/follow_rate.py 262144 10 >> gpurun_out/prof_b.log 2>&1
ls gpurun_out/r03_*_kernel_stats.csv
