# usage (GPU box): bash tools/timeline_round.sh <tag> [workload=code2] [mode=mixed] [extra bench args]
# rocprofv3 kernel trace of the clean bench loop -> kernel summary, per-step text timeline, and the per-stream busy / gap /
# critical-path JSON (tools/timeline_json.py); the trace db travels back (gzip) so the tools can be re-run off the box.
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; TAG=${1:-r05a}; W=${2:-code2}; M=${3:-mixed}; shift; shift; shift
O=gpurun_out/$TAG; mkdir -p $O
rm -rf /tmp/prof_${W}_$M
rocprofv3 --kernel-trace --stats -d /tmp/prof_${W}_$M -o res -- python bench.py --workload $W --mode $M --steps 30 --warmup 10 --no-cpu-baseline --no-kernel-timing --no-extra "$@" > $O/prof_${W}_$M.log 2>&1 || true
db=$(find /tmp/prof_${W}_$M -name "*.db" | head -1)
python tools/rocpd_summary.py $db 52 $O/${TAG}_${W}_${M} > /dev/null 2>&1 || true
python tools/timeline.py $db $O/${TAG}_${W}_${M}_timeline.txt 20 > /dev/null 2>&1 || true
python tools/timeline_json.py $db $O/${TAG}_timeline_${W}_${M}.json 24 > $O/${TAG}_timeline_${W}_${M}.txt 2>&1 || true
sz=$(stat -c %s $db); if [ $sz -lt 40000000 ]; then gzip -c $db > $O/${W}_${M}.db.gz; fi
tail -n 3 $O/prof_${W}_$M.log
head -n 12 $O/${TAG}_timeline_${W}_${M}.txt
