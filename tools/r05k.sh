cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r05k; O=gpurun_out/r05k
timeout 1200 python -m pytest tests/test_hip_linear3r.py tests/test_hip_linear.py tests/test_hip_engine.py tests/test_hip_configs.py tests/test_hip_dp.py tests/test_kernel_resources.py -x -q > $O/tests.log 2>&1; echo "tests rc $?" >> $O/tests.log
tail -n 12 $O/tests.log
run() { n=$1; shift; e=$1; shift
  env $e timeout 300 python bench.py --no-cpu-baseline --no-extra --no-kernel-timing --steps 100 "$@" > $O/$n.json 2>/dev/null
  python -c "
import json
d=json.load(open('$O/$n.json')); print('$n', d['value'], d['ms_per_step'], d.get('host_enqueue_ms_per_step'))"
}
for i in 1 2; do
run code2_fuse "GT_FUSE_BN=1"
run code2_nofuse "GT_FUSE_BN=0"
done
run er_fuse "GT_FUSE_BN=1" --workload er --steps 30
run er_nofuse "GT_FUSE_BN=0" --workload er --steps 30
run nci1_fuse "GT_FUSE_BN=1" --workload nci1
