cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r05q; O=gpurun_out/r05q
run() { n=$1; shift; e=$1; shift
  env $e timeout 300 python bench.py --no-cpu-baseline --no-extra --no-kernel-timing --steps 100 "$@" > $O/$n.json 2>$O/$n.err
  python -c "
import json
d=json.load(open('$O/$n.json')); print('$n', d['value'], d['ms_per_step'], d.get('host_enqueue_ms_per_step'))"
}
for i in 1 2; do
run part256 "GT_BN_MAX_PART=256"
run part512 "GT_BN_MAX_PART=512"
run part1024 "GT_BN_MAX_PART=1024"
run part128 "GT_BN_MAX_PART=128"
done
