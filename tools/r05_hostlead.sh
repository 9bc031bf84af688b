# usage (GPU box): bash tools/r05_hostlead.sh [workload] -- kernel trace + HIP runtime API trace of a short bench loop (no counters):
# how far ahead of the GPU is the host when it enqueues each kernel?  (tools/host_lead.py reads the db)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; W=${1:-code2}; O=gpurun_out/r05host; mkdir -p $O
rm -rf /tmp/prof_host
rocprofv3 --kernel-trace --hip-runtime-trace -d /tmp/prof_host -o res -- python bench.py --workload $W --steps 16 --warmup 8 --no-cpu-baseline --no-kernel-timing --no-extra > $O/prof.log 2>&1 || true
db=$(find /tmp/prof_host -name "*.db" | head -1)
ls -la $db
python tools/host_lead.py $db > $O/host_lead_$W.txt 2>&1 || true
head -60 $O/host_lead_$W.txt
sz=$(stat -c %s $db); if [ $sz -lt 60000000 ]; then gzip -c $db > $O/host_$W.db.gz; fi
