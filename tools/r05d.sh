cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r05d; O=gpurun_out/r05d
timeout 1200 python -m pytest tests/test_hip_linear3r.py tests/test_hip_linear3x.py tests/test_hip_engine.py tests/test_hip_configs.py tests/test_hip_parity.py tests/test_kernel_resources.py -x -q > $O/tests.log 2>&1; echo "tests rc $?" >> $O/tests.log
tail -n 5 $O/tests.log
for i in 1 2; do
timeout 300 python bench.py --no-cpu-baseline --no-extra --no-kernel-timing --steps 100 > $O/bench_code2_clean.json 2> $O/bench_code2_clean.err
GT_LIN3R=0 timeout 300 python bench.py --no-cpu-baseline --no-extra --no-kernel-timing --steps 100 > $O/bench_code2_clean_old.json 2>/dev/null
python -c "
import json
for f in ('bench_code2_clean','bench_code2_clean_old'):
    d=json.load(open('$O/'+f+'.json')); print(f, d['value'], d['ms_per_step'])"
done
timeout 300 python bench.py --no-cpu-baseline --no-extra --no-kernel-timing --steps 100 --mode fp32 > $O/bench_code2_fp32.json 2>/dev/null
timeout 300 python bench.py --no-cpu-baseline --no-extra --no-kernel-timing --steps 50 --workload er > $O/bench_er.json 2>/dev/null
GT_LIN3R=0 timeout 300 python bench.py --no-cpu-baseline --no-extra --no-kernel-timing --steps 50 --workload er > $O/bench_er_old.json 2>/dev/null
timeout 300 python bench.py --no-cpu-baseline --no-extra --no-kernel-timing --steps 100 --workload code2-pna > $O/bench_pna.json 2>/dev/null
GT_LIN3R=0 timeout 300 python bench.py --no-cpu-baseline --no-extra --no-kernel-timing --steps 100 --workload code2-pna > $O/bench_pna_old.json 2>/dev/null
python -c "
import json
for f in ('bench_code2_fp32','bench_er','bench_er_old','bench_pna','bench_pna_old'):
    d=json.load(open('$O/'+f+'.json')); print(f, d['value'], d['ms_per_step'])"
