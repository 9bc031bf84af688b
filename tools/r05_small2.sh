# usage (GPU box): bash tools/r05_small2.sh -- enqueue-bound batches: fewer event operations (held forks / piggy-backed finishes / one-launch BatchNorm)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r05small; o=gpurun_out/r05small
for cfg in "code2 --batch 32" "molpcba" "nci1"; do
  for sw in "GT_X=0" "GT_HOLD_FORKS=1" "GT_FORK_PIGGYBACK=1" "GT_HOLD_FORKS=1 GT_FORK_PIGGYBACK=1" "GT_BN_COOP=1" "GT_HOLD_FORKS=1 GT_BN_COOP=1" "GT_X=0"; do
    set -- $cfg
    w=$1; shift
    env $sw python bench.py --workload $w "$@" --steps 300 --warmup 30 --no-cpu-baseline --no-kernel-timing --no-extra 2>/dev/null | python -c "import sys,json;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print('$cfg | $sw |',d['value'],d['ms_per_step'],d.get('host_enqueue_ms_per_step'))"
  done
done | tee $o/event_ops_small.txt
