# usage (GPU box): bash tools/profile_quick.sh <tag>  -- bench lines + rocprofv3 kernel summaries of the two headline workloads (no PMC passes)
# (rocpd_summary divides by 52 steps: 10 warm-up + 30 timed + the 12 single steps of the idle-device host measurement)
TAG=${1:-r03a}
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/$TAG; mkdir -p $O
python bench.py --steps 50 --warmup 10 > $O/${TAG}_bench_code2.json 2> $O/bench_code2.err
python bench.py --workload molpcba --steps 50 --warmup 10 > $O/${TAG}_bench_molpcba.json 2> $O/bench_molpcba.err
for w in code2 molpcba; do
  rm -rf /tmp/prof_${w}
  rocprofv3 --kernel-trace --stats -d /tmp/prof_${w} -o res -- python bench.py --workload $w --steps 30 --warmup 10 --no-cpu-baseline --no-kernel-timing --no-extra > $O/prof_${w}.log 2>&1 || true
  db=$(find /tmp/prof_${w} -name "*.db" | head -1)
  python tools/rocpd_summary.py $db 52 $O/${TAG}_${w}_b256_mixed >> $O/prof_${w}.log 2>&1 || true
  python tools/timeline.py $db $O/${TAG}_${w}_timeline.txt 3 > /dev/null 2>&1 || true
done
ls $O
