set -e
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/marks2; mkdir -p $O
timeout -k 10 300 python -u tools/ggx_marks_probe.py 512 > $O/probe512.log 2>&1
cat $O/probe512.log
timeout -k 10 300 python -u -m pytest tests -m gpu -x -q --timeout 180 --timeout-method thread -k "marked or sun or sixteen or pins or ggx_level" > $O/pytest.log 2>&1
tail -3 $O/pytest.log
timeout -k 10 400 python -u tools/ggx_marks_probe.py 2048 > $O/probe2048.log 2>&1
cat $O/probe2048.log
