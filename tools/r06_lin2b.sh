# k_lin2 policy A/B on the ER step: lin_ring 0 (default) / 3 (only shapes k_lin1 does not cover) / 1 (never), two rounds
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r06l2; mkdir -p $O
for r in 1 2; do for v in 0 3 1; do
  python bench.py --workload er --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timing --no-extra --lib-option lin_ring=$v 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('lin_ring=$v', d['value'], d['ms_per_step'])"
done; done 2>&1 | tee $O/er_policy_ab.txt
bash tools/prof_one.sh r06l2 er --steps 20 --warmup 5 | head -30
