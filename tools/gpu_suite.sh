# usage (GPU box): bash tools/gpu_suite.sh <tag> [pytest args]  -- the full GPU suite + smoke, log under gpurun_out/<tag>/
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/${1:-r06}; mkdir -p $O; shift
python -m pytest tests -m gpu -q -x "$@" > $O/pytest_gpu.txt 2>&1; echo "pytest rc $?"
tail -15 $O/pytest_gpu.txt
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
