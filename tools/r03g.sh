cd $GRAFT_REPO_ROOT
O=gpurun_out/r03g; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.txt 2>&1
echo "pytest rc $?" >> $O/pytest_gpu.txt
timeout 300 python tools/gemm3_bench.py > $O/gemm3_bench.txt 2>&1
for w in code2 molpcba; do
  timeout 300 python bench.py --workload $w --no-cpu-baseline --no-extra --steps 50 --warmup 10 > $O/bench_$w.json 2> $O/bench_$w.err
  GT_F32_GEMM=exact timeout 300 python bench.py --workload $w --no-cpu-baseline --no-extra --steps 50 --warmup 10 > $O/bench_${w}_exact.json 2> $O/bench_${w}_exact.err
done
timeout 300 python bench.py --workload er --no-cpu-baseline --no-extra --steps 20 --warmup 5 > $O/bench_er.json 2> $O/bench_er.err
GT_F32_GEMM=exact timeout 300 python bench.py --workload er --no-cpu-baseline --no-extra --steps 20 --warmup 5 > $O/bench_er_exact.json 2> $O/bench_er_exact.err
timeout 300 python bench.py --workload code2 --mode fp32 --no-cpu-baseline --no-extra --steps 30 --warmup 10 > $O/bench_code2_fp32.json 2> $O/bench_code2_fp32.err
tail -5 $O/pytest_gpu.txt
