cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r03al; mkdir -p $O
timeout 1500 python -m pytest tests/test_hip_engine.py tests/test_hip_configs.py tests/test_hip_linear3x.py -x -q > $O/pytest.txt 2>&1; grep -E "passed|failed|error" $O/pytest.txt | tail -3
run() { tag=$1; w=$2; shift; shift; env "$@" python bench.py --workload $w --no-kernel-timing --no-cpu-baseline --no-extra --steps 100 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][0]); print('$tag', '$w', d['value'], d['ms_per_step'], d.get('host_enqueue_ms_per_step'), d['final_loss'])"; }
for rep in 1 2; do
run new code2 A=1
run old code2 GT_LIB_PATH=$GRAFT_REPO_ROOT/graphtrans_amd/libgt_old_attn.so
done
rm -rf /tmp/prof_code2
rocprofv3 --kernel-trace --stats -d /tmp/prof_code2 -o res -- python bench.py --workload code2 --steps 30 --warmup 10 --no-cpu-baseline --no-kernel-timing --no-extra > $O/prof_code2.log 2>&1 || true
db=$(find /tmp/prof_code2 -name "*.db" | head -1)
python tools/rocpd_summary.py $db 40 $O/r03al_code2_b256_mixed >> $O/prof_code2.log 2>&1 || true
python tools/timeline.py $db $O/r03al_code2_timeline.txt 3 > /dev/null 2>&1 || true
grep -E "k_small|k_lin3|k_linear_dx|launches per step" $O/r03al_code2_b256_mixed_summary.txt
