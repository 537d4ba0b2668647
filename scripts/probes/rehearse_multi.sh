#!/bin/bash
# bench.py's N > 1 code path on ONE card: two / four ranks over gloo, all on cuda:0 (the library's RCCL communicator refuses two
# ranks on one device, so the gather is the torch.distributed fallback: --allow-fallback-gather).  Not a measurement.
set -o pipefail
for N in 2 4; do
  for EXTRA in "" "--sorted-by-speed"; do
    echo "== ranks $N $EXTRA"
    TPC_BENCH_BACKEND=gloo TPC_BENCH_DEVICE=0 HSA_ENABLE_IPC_MODE_LEGACY=0 GLOO_SOCKET_IFNAME=lo timeout -k 10 300 \
      python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port $((29500 + N)) \
      bench.py --gpus $N --steps 3 --warmup 1 --batch 32768 --allow-fallback-gather $EXTRA 2> gpurun_out/rehearse_$N.err | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('value', round(d['value']), 'gather', d['gather'], 'n_gpus', d['n_gpus'], 'imbalance', d['imbalance'])
for r in d['ranks']: print('  ', r['rank'], r['shard'], r['iterations_total'], r['gather_verified'], round(r['pg_kernel_ms'],3))
" || { tail -5 gpurun_out/rehearse_$N.err; exit 1; }
  done
done
