exec < /dev/null
mkdir -p gpurun_out
for w in c2 c5s c4; do timeout 300 python bench.py --workload $w --warmup 1 2>gpurun_out/wl_$w.err | cut -c1-900; echo "rc=$? ($w)"; tail -2 gpurun_out/wl_$w.err | cut -c1-300; done
timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --workload c4 2>/dev/null | cut -c1-600
