P="bash scripts/profile.sh"
R=r03
$P ${R}_headline  python3 bench.py --steps 10 --warmup 2 --no-cpu --no-fp32 --no-pipelined --no-bit-exact --no-config2 --no-config5 > gpurun_out/prof_a.log 2>&1
$P ${R}_bitexact  python3 bench.py --steps 10 --warmup 2 --algo lane --no-cpu --no-fp32 --no-pipelined --no-config2 --no-config5 >> gpurun_out/prof_a.log 2>&1
$P ${R}_config2   python3 bench.py --steps 20 --warmup 2 --batch 4096 --horizon 10 --algo wave --no-cpu --no-fp32 --no-pipelined >> gpurun_out/prof_a.log 2>&1
$P ${R}_fp32      python3 bench.py --steps 10 --warmup 2 --dtype f32 --no-cpu --no-pipelined --no-config2 --no-config5 >> gpurun_out/prof_a.log 2>&1
ls gpurun_out/r03_*_kernel_stats.csv
python3 -c "import hashlib;print(hashlib.sha256(open('trajectory_controller_amd/lib/libtpc_mpc.so','rb').read()).hexdigest())"
