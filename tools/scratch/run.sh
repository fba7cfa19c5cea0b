set -e
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/wg2; mkdir -p $O
timeout -k 10 600 python -u -m pytest tests -m gpu -x -q --timeout 180 --timeout-method thread -k "ggx or bake or shard" > $O/pytest.log 2>&1
tail -2 $O/pytest.log
for wl in c3 c5; do
  steps=2; [ $wl = c3 ] && steps=20
  timeout -k 10 300 python bench.py --workload $wl --steps $steps --warmup 1 --no-cpu-baseline --no-extras 2>/dev/null \
      | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$wl', d['ms_per_step'], d['roofline']['kernel_ms'])" | tee -a $O/ab.log
done
