#!/bin/bash
# two / three ranks on one GPU (gloo), device-side exchange vs host recipe
cd /root/repo
export FQH_BENCH_BACKEND=gloo FQH_BENCH_ONE_GPU=1 MASTER_ADDR=127.0.0.1 HSA_ENABLE_IPC_MODE_LEGACY=0
for hp in 0 1; do
FQH_BENCH_HOST_PROTOCOL=$hp timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 2954$hp bench.py --gpus 2 --steps 10 --warmup 3 --bytes $((4<<30)) --shard-stats 2>&1 | grep "^{" | python -c "
import sys, json
for l in sys.stdin:
    j = json.loads(l)
    print({k: j[k] for k in ('value', 'ms_per_step', 'n_gpus') if k in j}, j.get('config', {}).get('exchange'), j.get('mode'), j.get('check'))
"
done
# one rank, world 1 over nccl: the device protocol with RCCL
