# intermediate kernel-level bench lines (default bench.py = kernel timing on every 16th step)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out/r03p
python bench.py > gpurun_out/r03p/bench_code2.json 2> gpurun_out/r03p/bench_code2.err
python bench.py --workload molpcba --no-extra --no-cpu-baseline > gpurun_out/r03p/bench_molpcba.json 2> gpurun_out/r03p/bench_molpcba.err
python bench.py --workload er --steps 20 --warmup 5 --no-extra --no-cpu-baseline > gpurun_out/r03p/bench_er.json 2> gpurun_out/r03p/bench_er.err
python bench.py --workload code2-pna --no-extra --no-cpu-baseline > gpurun_out/r03p/bench_pna.json 2> gpurun_out/r03p/bench_pna.err
tail -c 600 gpurun_out/r03p/bench_code2.err
