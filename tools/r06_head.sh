# round-6 re-entry check of HEAD: full GPU suite + smoke, then the default bench line as the driver runs it
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r06h; mkdir -p $O
bash tools/gpu_suite.sh r06h
( time python bench.py --report $O/report_default.json ) > $O/bench_default.json 2> $O/bench_default.err
tail -c 2500 $O/bench_default.json; tail -3 $O/bench_default.err
