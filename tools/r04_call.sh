#!/bin/bash
# GPU call 6 of round 4: persistent weight-stationary 32-channel ResBlock round: vocoder tests (incl. bitwise equality with the two-launch path), bench
exec < /dev/null
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_vocoder.py -m gpu -x -q -p no:cacheprovider > gpurun_out/r04f_tests.log 2>&1; echo "voc tests rc=$?"; grep -v "^Extension modules" gpurun_out/r04f_tests.log | tail -8
timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-side --out gpurun_out/r04f_bench_full.json > gpurun_out/r04f_bench.json 2> gpurun_out/r04f_bench.err; echo "bench rc=$?"; cat gpurun_out/r04f_bench.json | cut -c1-2600; grep "^\[bench [0-9]" gpurun_out/r04f_bench.err | tail -5
python - <<'P'
import json
d=json.load(open('gpurun_out/r04f_bench_full.json'))
print(json.dumps(d['kernels']['vocoder']['by_class'],indent=0))
P
