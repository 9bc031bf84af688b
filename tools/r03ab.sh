cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r03ab; mkdir -p $O
run() { tag=$1; shift; env "$@" python bench.py --workload code2 --no-kernel-timing --no-cpu-baseline --no-extra --steps 100 > $O/code2_$tag.json 2>$O/err_$tag.txt; env "$@" python bench.py --workload molpcba --no-kernel-timing --no-cpu-baseline --no-extra --steps 100 > $O/molpcba_$tag.json 2>>$O/err_$tag.txt; }
run base A=1
run devkernarg1 HIP_FORCE_DEV_KERNARG=1
run devkernarg0 HIP_FORCE_DEV_KERNARG=0
run hwq8 GPU_MAX_HW_QUEUES=8
run noint HSA_ENABLE_INTERRUPT=0
run base2 A=1
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r03ab/*.json')):
    try:
        d=json.loads([l for l in open(f) if l.startswith('{')][0]); print(f, d['value'], d['ms_per_step'], d.get('ms_per_step_median_device'), d.get('host_enqueue_ms_per_step'), d.get('final_loss'))
    except Exception as e: print(f, 'ERR', e)
PY
