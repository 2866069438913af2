#!/bin/bash
exec < /dev/null
TAG=${1:-si}
mkdir -p gpurun_out
timeout 400 python -m pytest tests/ -x -q -m gpu -p no:cacheprovider > gpurun_out/${TAG}_tests.log 2>&1
echo "tests rc=$?"; grep -v "^Extension modules" gpurun_out/${TAG}_tests.log | tail -5
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/${TAG}_smoke.log 2>&1; echo "smoke rc=$?"; tail -1 gpurun_out/${TAG}_smoke.log
timeout 300 python bench.py > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err; echo "bench rc=$?"; cut -c1-330 gpurun_out/${TAG}_bench.json
