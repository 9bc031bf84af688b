cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r03ag; mkdir -p $O
for rep in 1 2; do
for v in new old; do
  if [ $v = old ]; then export GT_LIB_PATH=$GRAFT_REPO_ROOT/graphtrans_amd/libgt_old_attn.so; else unset GT_LIB_PATH; fi
  echo "== $v $rep"; timeout 600 python tools/attn_bench.py 2>&1 | grep -E "dropout 0.3 \[longest|ER\), d256 h4 \[longest" | grep kernels
  python bench.py --workload code2 --no-kernel-timing --no-cpu-baseline --no-extra --steps 100 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][0]); print('code2', d['value'], d['ms_per_step'], d['final_loss'])"
done; done 2>&1 | tee $O/ab.txt
