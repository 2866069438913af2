python -m pytest tests/test_gpu_facade_parity.py -m gpu -q 2>&1 | tail -4
for i in 1 2 3; do python bench.py --workload c3f --steps 6 --warmup 1 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('c3f', round(d['ms_per_step'],2), d['prefill_batches_per_step'], d['vocoder_batches_per_step'])"; done
python bench.py --steps 6 --warmup 1 --no-side --no-cpu-baseline --no-profile-pass 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('headline ms_per_step', d['ms_per_step'])"
