cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r03au; mkdir -p $O
run() { tag=$1; w=$2; shift; shift; timeout 300 env "$@" python bench.py --workload $w --no-kernel-timing --no-cpu-baseline --no-extra --steps 100 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][0]); print('$tag', '$w', d['value'], d['ms_per_step'], d.get('host_enqueue_ms_per_step'), d['final_loss'])"; }
for rep in 1 2 3; do
run fork code2 A=1
run main code2 GT_LN_FINISH_MAIN=1
done
for rep in 1 2; do
run fork molpcba A=1
run main molpcba GT_LN_FINISH_MAIN=1
done
