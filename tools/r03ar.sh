cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r03ar; mkdir -p $O
for m in mixed bf16; do
  GT_CHECK_NOSYNC=1 GT_CHECK_ITERS=6000 GT_CHECK_MODE=$m timeout 900 python tools/engine_check_full.py 2>&1 | grep -v amdgpu.ids | tail -6 | tee -a $O/check_$m.txt
done
GT_CHECK_NOSYNC=1 GT_CHECK_DROPOUT=1 GT_CHECK_ITERS=3000 GT_CHECK_MODE=mixed timeout 900 python tools/engine_check_full.py 2>&1 | grep -v amdgpu.ids | tail -6 | tee -a $O/check_dropout.txt
