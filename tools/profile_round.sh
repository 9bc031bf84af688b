# usage (GPU box): bash tools/profile_round.sh <round tag>   -- PMC passes first (bench.py reads their JSON), then the
# bench lines of all workloads and the rocprofv3 kernel summaries; everything lands in gpurun_out/<tag>/
set -e
TAG=${1:-r01s}
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/$TAG
bash tools/pmc_round.sh $TAG > gpurun_out/$TAG/pmc.log 2>&1 || true
cp gpurun_out/$TAG/${TAG}_*_pmc_traffic.json profiles/ 2>/dev/null || true   # picked up by bench.py below
python bench.py > gpurun_out/$TAG/bench_code2.json 2> gpurun_out/$TAG/bench_code2.err
python bench.py --workload molpcba > gpurun_out/$TAG/bench_molpcba.json 2> gpurun_out/$TAG/bench_molpcba.err
python bench.py --workload nci1 > gpurun_out/$TAG/bench_nci1.json 2> gpurun_out/$TAG/bench_nci1.err
python bench.py --workload er --steps 20 --warmup 5 > gpurun_out/$TAG/bench_er.json 2> gpurun_out/$TAG/bench_er.err
python bench.py --workload code2-pna > gpurun_out/$TAG/bench_code2pna.json 2> gpurun_out/$TAG/bench_code2pna.err
python bench.py --no-kernel-timing --no-cpu-baseline --steps 100 > gpurun_out/$TAG/bench_code2_clean.json 2>/dev/null
python bench.py --no-kernel-timing --no-cpu-baseline --steps 100 --workload molpcba > gpurun_out/$TAG/bench_molpcba_clean.json 2>/dev/null
for w in code2 molpcba nci1; do
  rm -rf /tmp/prof_$w
  rocprofv3 --kernel-trace --stats -d /tmp/prof_$w -o res -- python bench.py --workload $w --steps 30 --warmup 10 --no-cpu-baseline --no-kernel-timing > gpurun_out/$TAG/prof_$w.log 2>&1 || true
  db=$(find /tmp/prof_$w -name "*.db" | head -1)
  python tools/rocpd_summary.py $db 40 gpurun_out/$TAG/${w} >> gpurun_out/$TAG/prof_$w.log 2>&1 || true
done
ls gpurun_out/$TAG
