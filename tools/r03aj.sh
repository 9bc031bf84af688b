cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r03aj; mkdir -p $O
run() { tag=$1; w=$2; shift; shift; env "$@" python bench.py --workload $w --no-kernel-timing --no-cpu-baseline --no-extra --steps 100 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][0]); print('$tag', '$w', d['value'], d['ms_per_step'], d.get('host_enqueue_ms_per_step'), d['final_loss'])"; }
for rep in 1 2; do
run off code2 GT_BN_ONE=0
run on32 code2 GT_BN_ONE_PART=32
run on48 code2 GT_BN_ONE_PART=48
run on64 code2 GT_BN_ONE_PART=64
run on96 code2 GT_BN_ONE_PART=96
done
for rep in 1 2; do
run off molpcba GT_BN_ONE=0
run on32 molpcba GT_BN_ONE_PART=32
run on64 molpcba GT_BN_ONE_PART=64
done
