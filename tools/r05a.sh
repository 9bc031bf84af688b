cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05a
python bench.py > gpurun_out/r05a/bench_code2.json 2> gpurun_out/r05a/bench_code2.err
bash tools/r05_timeline.sh r05a code2 mixed
bash tools/r05_timeline.sh r05a code2 fp32
