set -e
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/mp64; mkdir -p $O
bash tools/ab_chain.sh c3 tools/variants/libibl_mp64.so 20 > $O/ab_c3.log 2>&1
cat $O/ab_c3.log
IBL_HIP_LIB=tools/variants/libibl_mp64.so timeout -k 10 300 python -u tools/ggx_marks_probe.py 512 > $O/probe512.log 2>&1
cat $O/probe512.log
