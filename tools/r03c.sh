set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03c; mkdir -p $O
timeout 900 python -m pytest tests/test_hip_linear3x.py -x -q -s > $O/pytest_linear3x.txt 2>&1
timeout 300 python tools/gemm3_bench.py > $O/gemm3_bench.txt 2>&1
timeout 300 python tools/host_phases.py 8 200 code2 > $O/host_phases_b8.txt 2>&1
timeout 300 python bench.py --no-cpu-baseline --no-extra --steps 50 --warmup 10 > $O/bench_code2.json 2> $O/bench_code2.err
timeout 300 python bench.py --workload molpcba --no-cpu-baseline --no-extra --steps 50 --warmup 10 > $O/bench_molpcba.json 2> $O/bench_molpcba.err
tail -30 $O/pytest_linear3x.txt
