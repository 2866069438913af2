#!/bin/bash
exec < /dev/null
mkdir -p gpurun_out
T=gpurun_out/r02p
run() { tag=$1; shift; env $ENVV timeout 300 python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-throughput-mode "$@" > ${T}_$tag.json 2> ${T}_$tag.err; python - $tag <<'PY'
import json,sys
f="gpurun_out/r02p_%s.json"%sys.argv[1]
try:
    j=json.load(open(f)); b=j["breakdown_ms_per_step"]; r=j["roofline_second_kernel"]; print(sys.argv[1], "vocoder", round(b["vocoder"],2), "convs", round(b["vocoder_convs"],2), "frac", round(r["frac"],3))
except Exception as e: print(f, "ERR", e)
PY
}
ENVV="AUR_X=1" run base
ENVV="AUR_F16_PF_XH_MINC=256" run pf256
ENVV="AUR_F16_PF_XH_MINC=128" run pf128
ENVV="AUR_F16_PF_XH_MINC=64" run pf64
ENVV="AUR_F16_PF_XH_MINC=16" run pf16
AUR_F16_PF_XH_MINC=16 timeout 400 python -m pytest tests/test_gpu_vocoder.py -m gpu -q --tb=short -p no:cacheprovider > ${T}_voc.log 2>&1; echo "voc(pf16) rc=$?"; tail -3 ${T}_voc.log
