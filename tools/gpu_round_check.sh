#!/bin/bash
# The driver's round-end commands on one GPU box: full -m gpu suite, smoke(), default bench, rocprofv3 stats of the bench.
# usage (from the repo root, through gpurun): bash tools/gpu_round_check.sh <tag>   -> gpurun_out/<tag>_*, gpurun_out/prof_<tag>/
exec < /dev/null
TAG=${1:-final}
mkdir -p gpurun_out
timeout 600 python -m pytest tests/ -x -q -m gpu -p no:cacheprovider > gpurun_out/${TAG}_tests.log 2>&1
echo "tests rc=$?"; grep -v "^Extension modules" gpurun_out/${TAG}_tests.log | tail -6
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/${TAG}_smoke.log 2>&1; echo "smoke rc=$?"; tail -3 gpurun_out/${TAG}_smoke.log
timeout 400 python bench.py --out gpurun_out/${TAG}_bench_full.json > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err; echo "bench rc=$?"; cut -c1-400 gpurun_out/${TAG}_bench.json; tail -4 gpurun_out/${TAG}_bench.err
timeout 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --no-cpu-baseline --out gpurun_out/${TAG}_bench_torchrun_full.json > gpurun_out/${TAG}_bench_torchrun.json 2> gpurun_out/${TAG}_bench_torchrun.err; echo "torchrun bench rc=$?"; cut -c1-300 gpurun_out/${TAG}_bench_torchrun.json
PROF_TIMEOUT=240 bash tools/prof.sh $TAG 2>&1 | head -14
