cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out/r03s
rm -rf /tmp/prof_c2
rocprofv3 --kernel-trace --stats -d /tmp/prof_c2 -o res -- python bench.py --workload code2 --steps 30 --warmup 10 --no-cpu-baseline --no-kernel-timing --no-extra > gpurun_out/r03s/prof.log 2>&1
db=$(find /tmp/prof_c2 -name "*.db" | head -1)
python tools/rocpd_summary.py $db 60 gpurun_out/r03s/r03s_code2 >> gpurun_out/r03s/prof.log 2>&1
python tools/timeline.py $db gpurun_out/r03s/timeline.txt 3 >> gpurun_out/r03s/prof.log 2>&1
python -m pytest tests/test_hip_linear1.py -x -q 2>&1 | tail -2
python bench.py --workload er --steps 20 --warmup 5 --no-extra --no-cpu-baseline > gpurun_out/r03s/bench_er_w1.json 2>/dev/null
GT_BF16_GEMM=tiled python bench.py --workload er --steps 20 --warmup 5 --no-extra --no-cpu-baseline > gpurun_out/r03s/bench_er_tiled.json 2>/dev/null
ls gpurun_out/r03s
