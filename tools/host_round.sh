cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r04b; mkdir -p $O
for w in nci1 code2; do
  python tools/host_profile.py $w 60 > $O/cprofile_$w.txt 2>&1
  rm -rf /tmp/hip_$w
  rocprofv3 --hip-runtime-trace -d /tmp/hip_$w -o res -- python bench.py --workload $w --steps 30 --warmup 10 --no-cpu-baseline --no-kernel-timing --no-extra > $O/hip_$w.log 2>&1
  python tools/hip_api_stats.py $(find /tmp/hip_$w -name "*.db" | head -1) 40 > $O/hip_api_$w.txt 2>&1
done
tail -30 $O/hip_api_*.txt
