# full GPU suite, then the profile round (PMC passes, bench lines of all workloads, rocprofv3 kernel summaries)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
TAG=${1:-r04a}
mkdir -p gpurun_out/$TAG
python -m pytest tests -m gpu -x -q > gpurun_out/$TAG/pytest_gpu.txt 2>&1
grep -E "passed|failed" gpurun_out/$TAG/pytest_gpu.txt | tail -3
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/$TAG/smoke.txt 2>&1
tail -1 gpurun_out/$TAG/smoke.txt
bash tools/profile_round.sh $TAG > gpurun_out/$TAG/profile_round.log 2>&1
tail -3 gpurun_out/$TAG/profile_round.log
