# sweep of the whole-read pass launch shape: GC_TEST_LONG_ORDER (0 none, 1 longest first, 2 length-balanced waves) x GC_TEST_LONG_TEAM (lanes per wave)
for o in ${ORDERS:-0 1 2}; do for t in ${TEAMS:-0 8 4}; do
  if [ $t = 0 ]; then unset GC_TEST_LONG_TEAM; else export GC_TEST_LONG_TEAM=$t; fi
  GC_TEST_LONG_ORDER=$o python bench.py --no-cpu-baseline --steps 2 --warmup 1 2>&1 | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('order',$o,'team',$t, d['value'], d['ms_per_step'], 'long', d['stage_ms']['k_long_extend_all_rounds'])"
done; done
