# usage (GPU box): bash tools/r04_engine.sh <tag>  -- engine tests, then host phases of the four workloads
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/${1:-r04d}; mkdir -p $O
timeout 900 python -m pytest tests/test_hip_engine.py tests/test_hip_dp.py tests/test_hip_bn_sync.py -x -q > $O/pytest_engine.txt 2>&1; tail -n 25 $O/pytest_engine.txt
for w in code2 molpcba nci1; do timeout 300 python tools/host_phases.py $w 100 > $O/host_$w.txt 2>&1; tail -n 14 $O/host_$w.txt; done
