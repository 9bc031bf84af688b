# usage (GPU box): bash tools/r05_small.sh -- the stream forks at small batches: Code2 b32, Molpcba, NCI1 with / without the side streams
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r05small; o=gpurun_out/r05small
run() { # name, env..., -- args
  name=$1; shift
  env "$@" > /dev/null 2>&1
}
for cfg in "code2 --batch 32" "molpcba" "nci1"; do
  for sw in "GT_OVERLAP_VN=1 GT_OVERLAP_DW=1 GT_PREP_OVERLAP=1" "GT_OVERLAP_VN=0 GT_OVERLAP_DW=1 GT_PREP_OVERLAP=1" "GT_OVERLAP_VN=1 GT_OVERLAP_DW=0 GT_PREP_OVERLAP=1" "GT_OVERLAP_VN=0 GT_OVERLAP_DW=0 GT_PREP_OVERLAP=0" "GT_OVERLAP_VN=1 GT_OVERLAP_DW=1 GT_PREP_OVERLAP=0"; do
    set -- $cfg
    w=$1; shift
    env $sw python bench.py --workload $w "$@" --steps 200 --warmup 30 --no-cpu-baseline --no-kernel-timing --no-extra 2>/dev/null | python -c "import sys,json;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print('$cfg | $sw |',d['value'],d['ms_per_step'],d.get('host_enqueue_ms_per_step'))"
  done
done | tee $o/streams_small.txt
