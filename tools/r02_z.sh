#!/bin/bash
exec < /dev/null
mkdir -p gpurun_out
run() { env $ENVV timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-throughput-mode > gpurun_out/r02z.json 2> gpurun_out/r02z.err
  python - "$ENVV" <<'PY'
import json,sys
j=json.load(open("gpurun_out/r02z.json")); b=j["breakdown_ms_per_step"]
print(repr(sys.argv[1]), round(j["ms_per_step"],1), "prefill", round(b["gpt_prefill"],2), "decode step", round(b["gpt_ms_per_decode_step"],3))
PY
}
ENVV="AUR_GEMM_TILE_SMALL_N=1024" run
ENVV="AUR_GEMM_TILE_SMALL_N=4096" run
ENVV="AUR_GEMM_TILE_SMALL_N=0" run
