#!/bin/bash
# r03 experiment L: LDS-DMA staged, multi-buffered ResBlock convs
exec < /dev/null
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_vocoder.py -m gpu -x -q -s 2>&1 | grep -v "^$" | tail -25 > gpurun_out/r03_l_tests.log
echo "tests rc=$?"; tail -12 gpurun_out/r03_l_tests.log
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -Wno-unused-result tools/conv_diag.hip -o /tmp/conv_diag && timeout 300 /tmp/conv_diag > gpurun_out/r03_l_conv_diag.log 2>&1
cut -c1-60,150-260 gpurun_out/r03_l_conv_diag.log
for dma in 1 0; do
AUR_CONV_DMA=$dma timeout 600 python bench.py --no-cpu-baseline > gpurun_out/r03_l_bench_dma$dma.json 2> gpurun_out/r03_l_bench.err
python - $dma <<'PY'
import json,sys
d=json.loads(open('gpurun_out/r03_l_bench_dma%s.json'%sys.argv[1]).read().strip().splitlines()[-1])
print('dma',sys.argv[1],d['value'], d['ms_per_step'], json.dumps(d['breakdown_ms_per_step']))
rv=d['roofline_vocoder']
for c in rv['by_class']: print(c['class'],c['launches'],round(c['ms'],2),round(c['hbm']['frac'],3),round(c['mfma']['frac'],3))
PY
done
