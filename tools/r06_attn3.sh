# attention prefetch fix, same-box A/B against the previous attention.hip (graphtrans_amd/libgt_old.so): micro-bench + steps
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r06at; mkdir -p $O
for v in old new old new; do
  unset GT_LIB_PATH; [ $v = old ] && export GT_LIB_PATH=$PWD/graphtrans_amd/libgt_old.so
  echo "== $v"; python tools/attn_bench.py 2>/dev/null | grep -E "kernels" | grep -E "longest first" | grep -E "Code2-like batch, dropout|ER|one of 1001"
done 2>&1 | tee $O/attn_ab.txt
unset GT_LIB_PATH
bash tools/ab.sh 2>&1 | tee $O/ab_code2.txt
bash tools/ab.sh --workload er --steps 20 2>&1 | tee $O/ab_er.txt
bash tools/ab.sh --workload molpcba 2>&1 | tee $O/ab_molpcba.txt
