cd $GRAFT_REPO_ROOT
O=gpurun_out/r03i; mkdir -p $O
timeout 900 python -m pytest tests/test_hip_configs.py tests/test_hip_linear3x.py tests/test_hip_linear.py -q -x > $O/pytest.txt 2>&1
echo "rc $?" >> $O/pytest.txt
for mp in 256 512 1024; do
  GT_BN_MAX_PART=$mp timeout 300 python bench.py --no-cpu-baseline --no-extra --steps 60 --warmup 10 --no-kernel-timing > $O/bench_code2_mp$mp.json 2>/dev/null
  GT_BN_MAX_PART=$mp timeout 300 python bench.py --workload molpcba --no-cpu-baseline --no-extra --steps 60 --warmup 10 --no-kernel-timing > $O/bench_molpcba_mp$mp.json 2>/dev/null
done
timeout 300 python bench.py --no-cpu-baseline --no-extra --steps 50 --warmup 10 > $O/bench_code2_kt.json 2> $O/bench_code2_kt.err
tail -4 $O/pytest.txt
