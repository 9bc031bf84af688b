cd $GRAFT_REPO_ROOT
O=gpurun_out/r03m; mkdir -p $O
GT_BENCH_SHARE_GPU=1 GT_BENCH_BACKEND=gloo timeout 600 python bench.py --gpus 2 --steps 6 --warmup 2 --no-cpu-baseline > $O/bench_2rank_gloo.json 2> $O/bench_2rank_gloo.err
echo "rc $?" >> $O/bench_2rank_gloo.err
tail -c 1500 $O/bench_2rank_gloo.json
