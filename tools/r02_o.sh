#!/bin/bash
exec < /dev/null
mkdir -p gpurun_out
T=gpurun_out/r02o
timeout 400 python -m pytest tests/test_gpu_vocoder.py tests/test_gpu_kernels.py -m gpu -q --tb=short -p no:cacheprovider -k "voc or conv" > ${T}_voc.log 2>&1; echo "voc rc=$?"; tail -8 ${T}_voc.log
run() { tag=$1; shift; env $ENVV timeout 300 python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-throughput-mode "$@" > ${T}_$tag.json 2> ${T}_$tag.err; python - $tag <<'PY'
import json,sys
f="gpurun_out/r02o_%s.json"%sys.argv[1]
try:
    j=json.load(open(f)); b=j["breakdown_ms_per_step"]; r=j["roofline_second_kernel"]; print(sys.argv[1], "vocoder", round(b["vocoder"],2), "convs", round(b["vocoder_convs"],2), "frac", round(r["frac"],3), r.get("survey_8d_fp32_bytes"))
except Exception as e: print(f, "ERR", e)
PY
}
ENVV="AUR_VOC_ACT2=1" run act2
ENVV="AUR_VOC_ACT2=0" run noact2
timeout 600 python -m pytest tests/test_gpu_baseline_size.py -m gpu -q --tb=short -p no:cacheprovider -k "c2_greedy" > ${T}_c2.log 2>&1; echo "c2 rc=$?"; tail -3 ${T}_c2.log
