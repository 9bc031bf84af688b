cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r05p; O=gpurun_out/r05p
run() { n=$1; shift; e=$1; shift
  env $e timeout 300 python bench.py --no-cpu-baseline --no-extra --no-kernel-timing --steps 100 "$@" > $O/$n.json 2>$O/$n.err
  python -c "
import json
d=json.load(open('$O/$n.json')); print('$n', d['value'], d['ms_per_step'], d.get('host_enqueue_ms_per_step'))"
}
for i in 1 2 3; do run code2 "GT_X=1"; done
run molpcba "GT_X=1" --workload molpcba
timeout 900 python -m pytest tests/test_hip_engine.py tests/test_hip_parity.py tests/test_hip_segment.py -x -q > $O/tests.log 2>&1; echo "tests rc $?" >> $O/tests.log
tail -n 3 $O/tests.log
