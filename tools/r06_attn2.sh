# attention prefetch fix (branch-free TilePair, aux loads combined at store time): parity, micro-bench, Code2 / ER steps
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r06at; mkdir -p $O
timeout 1200 python -m pytest tests/test_hip_attention.py tests/test_hip_parity.py tests/test_hip_options.py -q -x > $O/pytest_attn.txt 2>&1; tail -5 $O/pytest_attn.txt
timeout 300 python tools/attn_bench.py > $O/attn_bench.txt 2>&1; tail -25 $O/attn_bench.txt
for w in code2 er; do S=100; [ $w = er ] && S=20
python bench.py --workload $w --steps $S --warmup 10 --no-cpu-baseline --no-kernel-timing --no-extra 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$w', d['value'], d['ms_per_step'])"
done 2>&1 | tee $O/steps.txt
python bench.py --workload code2 --steps 100 --warmup 10 --no-cpu-baseline --no-kernel-timing --no-extra --mode fp32 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('code2 fp32', d['value'], d['ms_per_step'])" | tee -a $O/steps.txt
bash tools/prof_one.sh r06at code2 | grep -E "attn|GPU kernel"
bash tools/prof_one.sh r06at er --steps 20 --warmup 5 | grep -E "attn|GPU kernel"
