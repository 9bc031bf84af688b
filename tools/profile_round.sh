# usage (GPU box): bash tools/profile_round.sh <round tag>   -- PMC passes first (bench.py attaches their build-stamped JSON),
# (rocpd_summary divides by 52 steps: 10 warm-up + 30 timed + the 12 single steps of the idle-device host measurement)
# then the bench lines of all workloads and the rocprofv3 kernel summaries; everything lands in gpurun_out/<tag>/
set -e
TAG=${1:-r02a}
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/$TAG
bash tools/pmc_round.sh $TAG mixed bf16 fp32 > gpurun_out/$TAG/pmc.log 2>&1 || true
cp gpurun_out/$TAG/${TAG}_*_pmc_traffic.json profiles/ 2>/dev/null || true   # picked up by bench.py below (same build id)
python bench.py --report gpurun_out/$TAG/report_code2.json > gpurun_out/$TAG/bench_code2.json 2> gpurun_out/$TAG/bench_code2.err
python bench.py --workload molpcba --report gpurun_out/$TAG/report_molpcba.json > gpurun_out/$TAG/bench_molpcba.json 2> gpurun_out/$TAG/bench_molpcba.err
python bench.py --workload nci1 --no-extra --report gpurun_out/$TAG/report_nci1.json > gpurun_out/$TAG/bench_nci1.json 2> gpurun_out/$TAG/bench_nci1.err
python bench.py --workload er --steps 20 --warmup 5 --no-extra --report gpurun_out/$TAG/report_er.json > gpurun_out/$TAG/bench_er.json 2> gpurun_out/$TAG/bench_er.err
python bench.py --workload er --steps 20 --warmup 5 --no-extra --mode bf16 --no-cpu-baseline --report gpurun_out/$TAG/report_er_bf16.json > gpurun_out/$TAG/bench_er_bf16.json 2> gpurun_out/$TAG/bench_er_bf16.err
python bench.py --workload code2-pna --no-extra --report gpurun_out/$TAG/report_code2pna.json > gpurun_out/$TAG/bench_code2pna.json 2> gpurun_out/$TAG/bench_code2pna.err
python bench.py --no-kernel-timing --no-cpu-baseline --no-extra --steps 100 --report gpurun_out/$TAG/report_code2_mixed_clean.json > gpurun_out/$TAG/bench_code2_mixed_clean.json 2>/dev/null
python bench.py --no-kernel-timing --no-cpu-baseline --no-extra --steps 100 --mode bf16 --report gpurun_out/$TAG/report_code2_bf16_clean.json > gpurun_out/$TAG/bench_code2_bf16_clean.json 2>/dev/null
# the per-rank shape of the global batch of 256 split over 8 GPUs (strong scaling), on one GPU
python bench.py --batch 32 --no-cpu-baseline --no-extra --report gpurun_out/$TAG/report_code2_b32.json > gpurun_out/$TAG/bench_code2_b32.json 2> gpurun_out/$TAG/bench_code2_b32.err
# clean lines (no kernel brackets, no CPU baseline) of the other workloads
for w in molpcba nci1 code2-pna er; do
  S=100; [ $w = er ] && S=20
  python bench.py --workload $w --steps $S --warmup 10 --no-cpu-baseline --no-kernel-timing --no-extra --report gpurun_out/$TAG/report_${w}_clean.json > gpurun_out/$TAG/bench_${w}_clean.json 2>/dev/null
done
python bench.py --no-kernel-timing --no-cpu-baseline --no-extra --steps 100 --mode fp32 --report gpurun_out/$TAG/report_code2_fp32_clean.json > gpurun_out/$TAG/bench_code2_fp32_clean.json 2>/dev/null
for m in mixed bf16 fp32; do
  for w in code2 molpcba er code2-pna nci1; do
    [ $w != code2 ] && [ $w != molpcba ] && [ $m != mixed ] && continue
    [ $w != code2 ] && [ $m = fp32 ] && continue
    rm -rf /tmp/prof_${w}_$m
    rocprofv3 --kernel-trace --stats -d /tmp/prof_${w}_$m -o res -- python bench.py --workload $w --mode $m --steps 30 --warmup 10 --no-cpu-baseline --no-kernel-timing --no-extra > gpurun_out/$TAG/prof_${w}_$m.log 2>&1 || true
    db=$(find /tmp/prof_${w}_$m -name "*.db" | head -1)
    python tools/rocpd_summary.py $db 52 gpurun_out/$TAG/${TAG}_${w}_b256_${m} >> gpurun_out/$TAG/prof_${w}_$m.log 2>&1 || true
    if [ $w = code2 ]; then   # per-stream busy / gap / critical-path record of the same trace
      python tools/timeline_json.py $db gpurun_out/$TAG/${TAG}_timeline_${w}_${m}.json 24 > gpurun_out/$TAG/${TAG}_timeline_${w}_${m}.txt 2>&1 || true
    fi
  done
done
ls gpurun_out/$TAG
