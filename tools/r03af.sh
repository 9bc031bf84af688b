cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r03af; mkdir -p $O
timeout 900 python -m pytest tests/test_hip_attention.py tests/test_hip_linear1.py -x -q > $O/pytest_attn.txt 2>&1; tail -5 $O/pytest_attn.txt
timeout 900 python -m pytest tests/test_hip_parity.py -x -q -k "G5 or G6 or G7 or G8 or pad or transformer or masked" > $O/pytest_par.txt 2>&1; tail -3 $O/pytest_par.txt
timeout 600 python tools/attn_bench.py 2>&1 | grep -v amdgpu.ids | grep -E "Code2-like|ER|256 x 126 " | tee $O/attn_bench.txt
for i in 1 2; do python bench.py --workload code2 --no-kernel-timing --no-cpu-baseline --no-extra --steps 100 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][0]); print('code2', d['value'], d['ms_per_step'], d['final_loss'])"; done
