# k_lin2 (csrc/linear2.h): parity tests, micro-bench at the ER and Code2 row counts, ER step
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r06l2; mkdir -p $O
timeout 900 python -m pytest tests/test_hip_linear2.py tests/test_hip_linear1.py tests/test_hip_options.py -q -x > $O/pytest_lin2.txt 2>&1; tail -15 $O/pytest_lin2.txt
timeout 300 python tools/gemm1_bench.py 131328 > $O/gemm1_bench_er.txt 2>&1; grep "M=" $O/gemm1_bench_er.txt
timeout 300 python tools/gemm1_bench.py 32000 > $O/gemm1_bench_code2.txt 2>&1; grep "M=" $O/gemm1_bench_code2.txt
timeout 600 python bench.py --workload er --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timing --no-extra > $O/bench_er_clean.json 2>$O/bench_er.err; cut -c1-260 $O/bench_er_clean.json; tail -3 $O/bench_er.err
