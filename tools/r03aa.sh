cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r03aa; mkdir -p $O
python bench.py --workload code2 --no-kernel-timing --no-cpu-baseline --no-extra --steps 100 > $O/bench_code2_clean.json 2>$O/bench.err
python bench.py --workload molpcba --no-kernel-timing --no-cpu-baseline --no-extra --steps 100 > $O/bench_molpcba_clean.json 2>>$O/bench.err
rm -rf /tmp/prof_code2
rocprofv3 --kernel-trace --stats -d /tmp/prof_code2 -o res -- python bench.py --workload code2 --steps 30 --warmup 10 --no-cpu-baseline --no-kernel-timing --no-extra > $O/prof_code2.log 2>&1 || true
db=$(find /tmp/prof_code2 -name "*.db" | head -1)
python tools/rocpd_summary.py $db 40 $O/r03aa_code2_b256_mixed >> $O/prof_code2.log 2>&1 || true
python tools/timeline.py $db $O/r03aa_code2_timeline.txt 3 > /dev/null 2>&1 || true
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r03aa/bench_*.json')):
    try:
        d=json.loads([l for l in open(f) if l.startswith('{')][0]); print(f, d['value'], d['ms_per_step'], d.get('ms_per_step_median_device'), d.get('host_enqueue_ms_per_step'), d.get('final_loss'))
    except Exception as e: print(f, 'ERR', e)
PY
