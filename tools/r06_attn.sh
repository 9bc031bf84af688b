# usage (GPU box): bash tools/r06_attn.sh <tag>  -- attention with 1 / 2 split groups: parity tests, the micro-benchmark, the Code2 / ER steps
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/${1:-r06d}; mkdir -p $O
timeout 900 python -m pytest tests/test_hip_options.py tests/test_hip_attention.py -m gpu -q -x > $O/pytest_attn.txt 2>&1; echo "pytest rc $?"; tail -4 $O/pytest_attn.txt
for g in 1 2; do timeout 300 python tools/attn_bench.py $g 2>&1 | grep -v "amdgpu.ids" | grep "kernels\|==" ; done | tee $O/attn_bench_split_groups.txt
for w in code2 er; do
  S=100; [ $w = er ] && S=20
  python bench.py --workload $w --steps $S --warmup 10 --no-cpu-baseline --no-kernel-timing --no-extra --report $O/report_${w}_clean.json 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$w', d['value'], d['ms_per_step'])"
done
