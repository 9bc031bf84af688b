# usage (GPU box): bash tools/r06_suite_bench.sh <tag>  -- full GPU suite, then clean bench lines (no brackets, no CPU leg)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/${1:-r06e}; mkdir -p $O
python -m pytest tests -m gpu -q -x > $O/pytest_gpu.txt 2>&1; echo "pytest rc $?"; tail -4 $O/pytest_gpu.txt
for w in "code2" "code2 --batch 32" "molpcba" "nci1" "code2-pna" "code2 --mode fp32"; do
  set -- $w; n=$(echo $w | tr -d ' -')
  S=100
  python bench.py --workload "$@" --steps $S --warmup 10 --no-cpu-baseline --no-kernel-timing --no-extra --report $O/report_${n}_clean.json 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$w', d['value'], d['ms_per_step'], d.get('host_enqueue_ms_per_step_idle_device'))"
done
