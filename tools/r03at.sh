cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r03at; mkdir -p $O
timeout 900 python -m pytest tests/test_hip_embed.py tests/test_hip_engine.py -x -q > $O/pytest.txt 2>&1; grep -E "passed|failed|error" $O/pytest.txt | tail -2
run() { tag=$1; w=$2; shift; shift; timeout 300 env "$@" python bench.py --workload $w --no-kernel-timing --no-cpu-baseline --no-extra --steps 100 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][0]); print('$tag', '$w', d['value'], d['ms_per_step'], d.get('host_enqueue_ms_per_step'), d['final_loss'])"; }
for rep in 1 2 3; do run new code2 A=1; done
run new molpcba A=1
rm -rf /tmp/prof_code2
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_code2 -o res -- python bench.py --workload code2 --steps 30 --warmup 10 --no-cpu-baseline --no-kernel-timing --no-extra > $O/prof_code2.log 2>&1 || true
db=$(find /tmp/prof_code2 -name "*.db" | head -1)
python tools/rocpd_summary.py $db 40 $O/r03at_code2_b256_mixed >> $O/prof_code2.log 2>&1 || true
grep -E "k_eseg|launches per step" $O/r03at_code2_b256_mixed_summary.txt
