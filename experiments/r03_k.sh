#!/bin/bash
# r03 experiment K: activated fp16 residual stream + fp16 MRF sum + fp16 transposed-conv output; non-temporal weight loads in gemm_rows
exec < /dev/null
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_vocoder.py -m gpu -x -q -s 2>&1 | grep -v "^$" | tail -25 > gpurun_out/r03_k_tests.log
echo "tests rc=$?"; tail -12 gpurun_out/r03_k_tests.log
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -Wno-unused-result tools/conv_diag.hip -o /tmp/conv_diag && timeout 300 /tmp/conv_diag > gpurun_out/r03_k_conv_diag.log 2>&1
cut -c1-150 gpurun_out/r03_k_conv_diag.log
for nt in 0 1; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -Wno-unused-result -DAUR_GEMM_W_NT=$nt tools/gemm_bench.hip -o /tmp/gemm_bench_$nt || exit 1
done
for rep in 1 2; do for nt in 0 1; do
  timeout 300 /tmp/gemm_bench_$nt 64 > gpurun_out/r03_k_gemm_bench_nt$nt.log 2>&1
  echo "nt=$nt"; grep "^chain" gpurun_out/r03_k_gemm_bench_nt$nt.log
done; done
timeout 600 python bench.py --no-cpu-baseline > gpurun_out/r03_k_bench.json 2> gpurun_out/r03_k_bench.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r03_k_bench.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], json.dumps(d['breakdown_ms_per_step']))
rv=d['roofline_vocoder']
print(rv['achieved'],rv['frac'],rv['avg_launch_ms'])
for c in rv['by_class']: print(c['class'],c['launches'],round(c['ms'],2),round(c['hbm']['frac'],3),round(c['mfma']['frac'],3))
PY
