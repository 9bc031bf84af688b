cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r03am; mkdir -p $O
run() { tag=$1; w=$2; shift; shift; env "$@" python bench.py --workload $w --no-kernel-timing --no-cpu-baseline --no-extra --steps 100 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][0]); print('$tag', '$w', d['value'], d['ms_per_step'], d.get('host_enqueue_ms_per_step'), d['final_loss'])"; }
for rep in 1 2 3; do
run dxfirst code2 A=1
run dwfirst code2 GT_HEADS_DW_FIRST=1
done
