# attention dK/dV: per-query statistics as four 16-byte LDS reads; A/B against the previous build (libgt_old.so)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r06at; mkdir -p $O
timeout 1200 python -m pytest tests/test_hip_attention.py -q -x > $O/pytest_attn5.txt 2>&1; tail -2 $O/pytest_attn5.txt
for v in old new old new; do
  unset GT_LIB_PATH; [ $v = old ] && export GT_LIB_PATH=$PWD/graphtrans_amd/libgt_old.so
  echo "== $v"; python tools/attn_bench.py 2>/dev/null | grep -E "kernels" | grep -E "longest first" | grep -E "Code2-like batch, dropout|ER|one of 1001"
done 2>&1 | tee $O/attn_ab5.txt
