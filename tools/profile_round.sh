set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r01p
# official bench lines (with kernel timing + cpu baseline)
python bench.py > gpurun_out/r01p/bench_code2.json 2> gpurun_out/r01p/bench_code2.err
python bench.py --workload molpcba > gpurun_out/r01p/bench_molpcba.json 2> gpurun_out/r01p/bench_molpcba.err
python bench.py --workload nci1 > gpurun_out/r01p/bench_nci1.json 2> gpurun_out/r01p/bench_nci1.err
for w in code2 molpcba nci1; do
  rm -rf /tmp/prof_$w
  rocprofv3 --kernel-trace --stats -d /tmp/prof_$w -o res -- python bench.py --workload $w --steps 30 --warmup 10 --no-cpu-baseline --no-kernel-timing > gpurun_out/r01p/prof_$w.log 2>&1 || true
  db=$(find /tmp/prof_$w -name "*.db" | head -1)
  python tools/rocpd_summary.py $db 40 gpurun_out/r01p/${w} >> gpurun_out/r01p/prof_$w.log 2>&1 || true
done
ls gpurun_out/r01p
