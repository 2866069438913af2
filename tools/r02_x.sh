#!/bin/bash
exec < /dev/null
mkdir -p gpurun_out
run() { env $ENVV timeout 300 python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-throughput-mode $1 > gpurun_out/r02x.json 2> gpurun_out/r02x.err
  python - "$ENVV $1" <<'PY'
import json,sys
j=json.load(open("gpurun_out/r02x.json")); b=j["breakdown_ms_per_step"]
print(repr(sys.argv[1]), round(j["ms_per_step"],1), {k:round(v,2) for k,v in b.items() if k in ("gpt","vocoder","gpt_ms_per_decode_step")})
PY
}
ENVV="AUR_VOC_LDS_PAD=0" run ""
ENVV="AUR_VOC_LDS_PAD=40000" run ""
ENVV="AUR_VOC_LDS_PAD=40000" run "--pipeline"
ENVV="AUR_VOC_LDS_PAD=16000" run "--pipeline"
