cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for v in 1 0; do
  rm -rf /tmp/prof_$v; AUR_SMALL_M=$v rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$v -o c2 -- python $R/bench.py --workload c2 --warmup 1 > /tmp/prof_$v.log 2>&1
  f=$(find /tmp/prof_$v -name "*kernel_stats.csv" | head -1)
  echo "== AUR_SMALL_M=$v  $(tail -1 /tmp/prof_$v.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('tta', d['time_to_audio_ms'], 'step', d['decode_step']['ms'])")"
  head -22 $f | cut -c1-200
  cp $f $R/gpurun_out/r06e_c2_small${v}_kernel_stats.csv
done
AUR_BENCH_STEP_TRACE=1 AUR_SMALL_M=1 python $R/bench.py --workload c2 --warmup 1 2>&1 | grep -i "step trace" | tail -3
