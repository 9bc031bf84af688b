# usage (GPU box): bash tools/r06_first.sh <tag> -- the driver's default bench command, then the line's size / keys
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/${1:-r06a}; mkdir -p $O
t0=$(date +%s); python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_default.out 2> $O/bench_default.err; rc=$?; t1=$(date +%s)
echo "rc $rc wall $((t1-t0)) s"; cp bench_report.json $O/bench_report.json
python - <<PY
import json
line=open("$O/bench_default.out").read().strip().splitlines()[-1]
print("line bytes", len(line)); d=json.loads(line); print(line)
PY
tail -5 $O/bench_default.err
