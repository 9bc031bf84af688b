# usage (GPU box): bash tools/r06_probe1.sh <tag>
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/${1:-r06c}; mkdir -p $O
python -m pytest tests/test_hip_options.py tests/test_hip_engine.py tests/test_hip_dp.py tests/test_hip_segment.py -m gpu -q -x > $O/pytest_part.txt 2>&1; echo "pytest rc $?"; tail -3 $O/pytest_part.txt
bash tools/launch_floor.sh $1 2>&1 | tee $O/launch_chain_floor.txt
for td in default 0; do
  [ $td = 0 ] && export GT_BENCH_TDROP=0
  python bench.py --steps 72 --warmup 10 --no-cpu-baseline --no-extra --report $O/report_tdrop_$td.json > /dev/null 2>&1
  python - <<PY
import json
d=json.load(open("$O/report_tdrop_$td.json")); k=d["kernels"]
print("tdrop=$td", d["value"], d["ms_per_step"], {n:(k[n]["avg_us"],k[n]["calls"]) for n in k if "attn" in n or "lin1" in n})
PY
done
