# usage (GPU box): bash tools/launch_floor.sh <tag>   -- the floor of the launch chain: the what-if library of
# tools/skip_probe_build.sh with GT_EMPTY=1 replaces every kernel by one empty 64-thread block on the SAME stream, so the step keeps
# its stream / event topology (three streams, ~75 fork / join pairs) and loses all its work.  Beside it: the real step, and the same
# chain with the side streams folded into the main one (GT_OVERLAP_DW=0 GT_OVERLAP_VN=0 GT_PREP_OVERLAP=0: no forks at all).
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/${1:-r06}; mkdir -p $O
export GT_LIB_PATH=$PWD/graphtrans_amd/libgt_skip.so
run() { n=$1; shift; e=$1; shift
  env $e python bench.py --steps 100 --warmup 10 --no-kernel-timing --no-cpu-baseline --no-extra --report $O/floor_$n.json "$@" 2>/dev/null | tail -1 | \
    python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%-34s %.4f ms/step  host idle-device enqueue %.3f ms' % ('$n', d['ms_per_step'], d.get('host_enqueue_ms_per_step_idle_device') or -1))"
}
for w in "code2 --workload code2" "code2_b32 --workload code2 --batch 32" "molpcba --workload molpcba" "nci1 --workload nci1"; do
  set -- $w; n=$1; shift
  run ${n}_real GT_X=0 "$@"
  run ${n}_empty GT_EMPTY=1 "$@"
  run ${n}_empty_one_stream "GT_EMPTY=1 GT_OVERLAP_DW=0 GT_OVERLAP_VN=0 GT_PREP_OVERLAP=0" "$@"
  run ${n}_real_one_stream "GT_OVERLAP_DW=0 GT_OVERLAP_VN=0 GT_PREP_OVERLAP=0" "$@"
done
