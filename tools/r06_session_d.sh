python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
for v in 1 0 1 0; do echo "== AUR_SMALL_M=$v"; AUR_SMALL_M=$v python bench.py --workload c2 --warmup 1 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print({k: d[k] for k in ('time_to_audio_ms','prefill_ms','decode_ms','vocoder_ms')}, d['decode_step']['ms'])"; done > gpurun_out/r06d_c2_ab.log 2>&1; cat gpurun_out/r06d_c2_ab.log
python -m pytest tests -m gpu -q 2>&1 | tail -40 > gpurun_out/r06d_gpu_tests.log; cat gpurun_out/r06d_gpu_tests.log
