cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r05g; O=gpurun_out/r05g
run() { n=$1; shift; e=$1; shift
  env $e timeout 300 python bench.py --no-cpu-baseline --no-extra --no-kernel-timing --steps 100 "$@" > $O/$n.json 2>/dev/null
  python -c "
import json
d=json.load(open('$O/$n.json')); print('$n', d['value'], d['ms_per_step'], d.get('host_enqueue_ms_per_step'))"
}
for i in 1 2; do
run base "GT_PRIO_VN=-1 GT_PRIO_DW=1"
run vn0 "GT_PRIO_VN=0 GT_PRIO_DW=1"
run vn0dw0 "GT_PRIO_VN=0 GT_PRIO_DW=0"
run vn1dw1 "GT_PRIO_VN=1 GT_PRIO_DW=1"
run lin3r_off "GT_LIN3R=0"
done
