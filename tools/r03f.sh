cd $GRAFT_REPO_ROOT
O=gpurun_out/r03f; mkdir -p $O
timeout 900 python -m pytest tests/test_hip_linear3x.py -x -q > $O/pytest_linear3x.txt 2>&1
for v in base mt4wb1 mt2wb1 base_noload base_nodma base_noa; do echo "== $v"; timeout 120 ./tools/gemm3_probe_$v; done > $O/probe.txt 2>&1
tail -3 $O/pytest_linear3x.txt; cat $O/probe.txt
