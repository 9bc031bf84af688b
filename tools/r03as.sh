cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r03as; mkdir -p $O
for m in mixed bf16; do
echo "== current $m"; GT_CHECK_ITERS=2 GT_CHECK_MODE=$m timeout 600 python tools/engine_check_full.py 2>&1 | grep "module path"
echo "== GT_DW16=0 $m"; GT_DW16=0 GT_CHECK_ITERS=2 GT_CHECK_MODE=$m timeout 600 python tools/engine_check_full.py 2>&1 | grep "module path"
echo "== old attention + GT_DW16=0 $m"; GT_LIB_PATH=$GRAFT_REPO_ROOT/graphtrans_amd/libgt_old_attn.so GT_DW16=0 GT_CHECK_ITERS=2 GT_CHECK_MODE=$m timeout 600 python tools/engine_check_full.py 2>&1 | grep "module path"
done 2>&1 | tee $O/cmp.txt
