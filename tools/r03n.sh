cd $GRAFT_REPO_ROOT
O=gpurun_out/r03n; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q -x > $O/pytest_gpu.txt 2>&1
echo "rc $?" >> $O/pytest_gpu.txt
for w in code2 molpcba nci1; do timeout 300 python bench.py --workload $w --no-cpu-baseline --no-extra --steps 60 --warmup 10 --no-kernel-timing > $O/bench_$w.json 2> $O/bench_$w.err; done
timeout 300 python bench.py --workload code2-pna --no-cpu-baseline --no-extra --steps 40 --warmup 10 --no-kernel-timing > $O/bench_pna.json 2> $O/bench_pna.err
grep -E "passed|failed" $O/pytest_gpu.txt | tail -2
