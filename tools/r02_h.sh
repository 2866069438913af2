#!/bin/bash
# vocoder staging A/B: synchronous vs software-pipelined (by input-channel threshold), 256- vs 512-position tiles
exec < /dev/null
mkdir -p gpurun_out
T=gpurun_out/r02h
run() { tag=$1; shift; env $ENVV timeout 300 python bench.py --steps 1 --warmup 0 --no-cpu-baseline "$@" > ${T}_$tag.json 2> ${T}_$tag.err; python - $tag <<'PY'
import json,sys
f="gpurun_out/r02h_%s.json"%sys.argv[1]
try:
    j=json.load(open(f)); b=j["breakdown_ms_per_step"]; print(sys.argv[1], "vocoder", round(b["vocoder"],2), "convs", round(b["vocoder_convs"],2), "step", round(j["ms_per_step"],1))
except Exception as e: print(f, "ERR", e)
PY
}
ENVV="AUR_X=0" run base
ENVV="AUR_F16_PF_MINC=256" run pf256
ENVV="AUR_F16_PF_MINC=128" run pf128
ENVV="AUR_F16_PF_MINC=64" run pf64
ENVV="AUR_F16_PF_MINC=16" run pf16
ENVV="AUR_F16_PF_MINC=256 AUR_F16_WIDE_MINC=256" run pf256w
ENVV="AUR_F16_PF_MINC=128 AUR_F16_WIDE_MINC=128" run pf128w
ENVV="AUR_F16_PF_MINC=64 AUR_F16_WIDE_MINC=64" run pf64w
ENVV="AUR_F16_PF_MINC=64 AUR_F16_WIDE_MINC=128" run pf64w128
timeout 300 python -m pytest tests/test_gpu_vocoder.py -m gpu -q --tb=short -p no:cacheprovider > ${T}_voc.log 2>&1; echo "voc tests (default) rc=$?"; tail -3 ${T}_voc.log
AUR_F16_PF_MINC=16 AUR_F16_WIDE_MINC=16 timeout 300 python -m pytest tests/test_gpu_vocoder.py tests/test_gpu_kernels.py -k "voc or conv" -m gpu -q --tb=short -p no:cacheprovider > ${T}_voc_pf.log 2>&1; echo "voc tests (pf+wide) rc=$?"; tail -3 ${T}_voc_pf.log
