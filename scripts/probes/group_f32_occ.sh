set -e
mkdir -p gpurun_out
for V in base f32occ2 f32occ3 f32occ4; do
  if [ $V = base ]; then unset TPC_MPC_LIB; else export TPC_MPC_LIB=$PWD/ab/$V/libtpc_mpc.so; fi
  echo "== $V" 
  python scripts/group_probe.py f32 20 2,4,8 16384,65536,131072,262144,524288
  python scripts/group_probe.py f32 10 2,4 65536,262144
  python scripts/group_probe.py f32 30 2,4 65536,262144
  python scripts/group_probe.py f32 40 2,4 65536,262144
done
