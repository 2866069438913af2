#!/bin/bash
exec < /dev/null
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_vocoder.py tests/test_gpu_api.py tests/test_conditioning.py -m gpu -x -q 2>&1 | tail -3
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-throughput-mode 2>/dev/null | cut -c1-260
