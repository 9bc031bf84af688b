# usage (GPU box): bash tools/r04_engine.sh <tag>  -- engine tests, then the clean bench lines
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/${1:-r04d}; mkdir -p $O
timeout 900 python -m pytest tests/test_hip_engine.py tests/test_hip_dp.py tests/test_hip_bn_sync.py tests/test_hip_configs.py -x -q > $O/pytest_engine.txt 2>&1; tail -n 5 $O/pytest_engine.txt
bash tools/bench_all.sh ${1:-r04d}
for b in 32; do python bench.py --workload code2 --batch $b --steps 200 --warmup 20 --no-cpu-baseline --no-kernel-timing --no-extra 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('code2 b$b', d['value'], d['ms_per_step'], 'host', d['host_enqueue_ms_per_step'])"; done
