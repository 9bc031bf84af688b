cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r05n; O=gpurun_out/r05n
run() { n=$1; shift; e=$1; shift
  env $e timeout 300 python bench.py --no-cpu-baseline --no-extra --no-kernel-timing --steps 100 "$@" > $O/$n.json 2>$O/$n.err
  python -c "
import json
d=json.load(open('$O/$n.json')); print('$n', d['value'], d['ms_per_step'], d.get('host_enqueue_ms_per_step'))"
}
for i in 1 2 3; do
run code2_piggy "GT_FORK_PIGGYBACK=1"
run code2_nopiggy "GT_FORK_PIGGYBACK=0"
done
run fp32_piggy "GT_FORK_PIGGYBACK=1" --mode fp32
run fp32_nopiggy "GT_FORK_PIGGYBACK=0" --mode fp32
run molpcba_piggy "GT_FORK_PIGGYBACK=1" --workload molpcba
run molpcba_nopiggy "GT_FORK_PIGGYBACK=0" --workload molpcba
timeout 1500 python -m pytest tests/test_hip_engine.py tests/test_hip_configs.py tests/test_hip_norm.py tests/test_hip_dp.py -x -q > $O/tests.log 2>&1; echo "tests rc $?" >> $O/tests.log
tail -n 4 $O/tests.log
