cd $GRAFT_REPO_ROOT
O=gpurun_out/r03l; mkdir -p $O
timeout 900 python -m pytest tests/test_hip_linear3x.py -q -x -s > $O/pytest_l3.txt 2>&1
echo "rc $?" >> $O/pytest_l3.txt
grep -E "^dW|passed|failed|Error|assert" $O/pytest_l3.txt | head -30
timeout 900 python -m pytest tests/test_hip_configs.py tests/test_hip_engine.py -q -x > $O/pytest_cfg.txt 2>&1
tail -3 $O/pytest_cfg.txt
for w in code2 molpcba; do timeout 300 python bench.py --workload $w --no-cpu-baseline --no-extra --steps 60 --warmup 10 > $O/bench_$w.json 2> $O/bench_$w.err; done
timeout 300 python bench.py --workload er --no-cpu-baseline --no-extra --steps 20 --warmup 5 > $O/bench_er.json 2> $O/bench_er.err
