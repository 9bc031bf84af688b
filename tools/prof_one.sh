# usage (GPU box): bash tools/prof_one.sh <tag> <workload> [extra bench args]  -- rocprofv3 kernel summary + timeline of one workload
# (rocpd_summary divides by 52 steps: 10 warm-up + 30 timed + the 12 single steps of the idle-device host measurement)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; TAG=${1:-r04}; W=${2:-code2}; shift; shift
O=gpurun_out/$TAG; mkdir -p $O
rm -rf /tmp/prof_$W
rocprofv3 --kernel-trace --stats -d /tmp/prof_$W -o res -- python bench.py --workload $W --steps 30 --warmup 10 --no-cpu-baseline --no-kernel-timing --no-extra "$@" > $O/prof_$W.log 2>&1 || true
db=$(find /tmp/prof_$W -name "*.db" | head -1)
python tools/rocpd_summary.py $db 52 $O/${TAG}_${W} > /dev/null 2>&1 || true
python tools/timeline.py $db $O/${TAG}_${W}_timeline.txt 3 > /dev/null 2>&1 || true
head -n 45 $O/${TAG}_${W}_summary.txt
