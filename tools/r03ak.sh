cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r03ak; mkdir -p $O
rm -rf /tmp/prof_code2
GT_BN_ONE_PART=64 rocprofv3 --kernel-trace --stats -d /tmp/prof_code2 -o res -- python bench.py --workload code2 --steps 30 --warmup 10 --no-cpu-baseline --no-kernel-timing --no-extra > $O/prof_code2.log 2>&1 || true
db=$(find /tmp/prof_code2 -name "*.db" | head -1)
python tools/rocpd_summary.py $db 40 $O/r03ak_code2_b256_mixed >> $O/prof_code2.log 2>&1 || true
python tools/timeline.py $db $O/r03ak_code2_timeline.txt 3 > /dev/null 2>&1 || true
grep -E "k_bn|launches per step" $O/r03ak_code2_b256_mixed_summary.txt
