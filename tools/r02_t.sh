#!/bin/bash
# GPU: in-library RCCL (world 1) tests, torchrun bench with the native broadcast, quick bench sanity
exec < /dev/null
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_edges.py -m gpu -x -q -k "rccl" 2>&1 | tail -8
AUR_NATIVE_BCAST=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --steps 1 --warmup 1 > gpurun_out/r02t_native_bcast.json 2> gpurun_out/r02t_native_bcast.err
echo "torchrun native bcast rc=$?"; tail -3 gpurun_out/r02t_native_bcast.err; cut -c1-300 gpurun_out/r02t_native_bcast.json
