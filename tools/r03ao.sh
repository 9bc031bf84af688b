cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r03ao; mkdir -p $O
run() { tag=$1; w=$2; shift; shift; env "$@" python bench.py --workload $w --no-kernel-timing --no-cpu-baseline --no-extra --steps 100 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][0]); print('$tag', '$w', d['value'], d['ms_per_step'], d.get('host_enqueue_ms_per_step'), d['final_loss'])"; }
for w in code2 molpcba; do
run base $w A=1
run optflush0 $w AMD_OPT_FLUSH=0
run sysscope0 $w ROC_SYSTEM_SCOPE_SIGNAL=0
run skipargcopy $w ROC_SKIP_KERNEL_ARG_COPY=1
run fgskernarg0 $w ROC_USE_FGS_KERNARG=0
run cpuwait $w ROC_CPU_WAIT_FOR_SIGNAL=0
run sdma0 $w HSA_ENABLE_SDMA=0
run base2 $w A=1
done 2>&1 | tee $O/env.txt
