cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r05b; O=gpurun_out/r05b
timeout 600 python -m pytest tests/test_hip_linear3r.py tests/test_hip_linear3x.py -x -q -s > $O/test_lin3r.log 2>&1; echo "tests rc $?" >> $O/test_lin3r.log
tail -n 5 $O/test_lin3r.log
timeout 300 python tools/gemm3r_bench.py > $O/gemm3r_bench.txt 2>&1
cat $O/gemm3r_bench.txt
timeout 300 python bench.py --no-cpu-baseline --no-extra --no-kernel-timing --steps 100 > $O/bench_code2_clean.json 2> $O/bench_code2_clean.err
GT_LIN3R=0 timeout 300 python bench.py --no-cpu-baseline --no-extra --no-kernel-timing --steps 100 > $O/bench_code2_clean_old.json 2>/dev/null
python -c "
import json
for f in ('bench_code2_clean','bench_code2_clean_old'):
    d=json.load(open('$O/'+f+'.json')); print(f, d['value'], d['ms_per_step'])"
