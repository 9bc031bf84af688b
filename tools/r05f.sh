cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r05f; O=gpurun_out/r05f
run() { # name, env, args
  n=$1; shift; e=$1; shift
  env $e timeout 300 python bench.py --no-cpu-baseline --no-extra --no-kernel-timing --steps 100 "$@" > $O/$n.json 2>/dev/null
  python -c "
import json
d=json.load(open('$O/$n.json')); print('$n', d['value'], d['ms_per_step'], d.get('host_enqueue_ms_per_step'))"
}
for i in 1 2; do
run molpcba_coop GT_BN_COOP=1 --workload molpcba
run molpcba_three GT_BN_COOP=0 --workload molpcba
run code2b32_coop GT_BN_COOP=1 --batch 32
run code2b32_three GT_BN_COOP=0 --batch 32
done
run code2_coop1 GT_BN_COOP=1
run code2_coop4 "GT_BN_COOP=1 GT_BN_COOP_ROUNDS=4"
run code2_three GT_BN_COOP=0
run pna_coop GT_BN_COOP=1 --workload code2-pna
run pna_three GT_BN_COOP=0 --workload code2-pna
