cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r03ae; mkdir -p $O
run() { tag=$1; shift; env "$@" python bench.py --workload code2 --no-kernel-timing --no-cpu-baseline --no-extra --steps 100 > $O/code2_$tag.json 2>$O/err_$tag.txt; }
run old1 GT_DW16=0
run new256a GT_DW16=1
run new128a GT_DW16_BLOCKS=128
run old2 GT_DW16=0
run new256b GT_DW16=1
run new192 GT_DW16_BLOCKS=192
run new128b GT_DW16_BLOCKS=128
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r03ae/*.json')):
    try:
        d=json.loads([l for l in open(f) if l.startswith('{')][0]); print(f, d['value'], d['ms_per_step'], d.get('ms_per_step_median_device'), d.get('host_enqueue_ms_per_step'), d.get('final_loss'))
    except Exception as e: print(f, 'ERR', e)
PY
