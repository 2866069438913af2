#!/bin/bash
# One GPU session: kernel + vocoder parity tests, vocoder A/B timing, bench, rocprof trace.  usage: tools/session.sh <tag>
exec < /dev/null
TAG=${1:-s}
mkdir -p gpurun_out
timeout 400 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_vocoder.py -m gpu -x -q -p no:cacheprovider > gpurun_out/${TAG}_tests.log 2>&1
echo "tests rc=$?"; grep -v "^Extension modules" gpurun_out/${TAG}_tests.log | tail -15
timeout 150 python tools/quick_perf.py 2 64 8 > gpurun_out/${TAG}_qp1.log 2>&1; echo "qp1 rc=$?"; grep vocode_wall gpurun_out/${TAG}_qp1.log | tail -1
AUR_XT_F16=0 timeout 150 python tools/quick_perf.py 2 64 8 > gpurun_out/${TAG}_qp0.log 2>&1; echo "qp0 rc=$?"; grep vocode_wall gpurun_out/${TAG}_qp0.log | tail -1
timeout 300 python bench.py --no-cpu-baseline > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err; echo "bench rc=$?"; cut -c1-600 gpurun_out/${TAG}_bench.json
PROF_TIMEOUT=240 bash tools/prof.sh $TAG 2>&1 | head -45
