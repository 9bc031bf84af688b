cd $GRAFT_REPO_ROOT
O=gpurun_out/r03d; mkdir -p $O
timeout 900 python -m pytest tests/test_hip_linear3x.py -x -q > $O/pytest_linear3x.txt 2>&1
for v in base mt4wb1 mt4wb2 mt2wb1 base_nodma base_noa base_noload base_nomfma base_nost mt4_nodma mt4_nomfma; do echo "== $v"; timeout 120 ./tools/gemm3_probe_$v; done > $O/probe.txt 2>&1
timeout 300 python tools/gemm3_bench.py > $O/gemm3_bench.txt 2>&1
tail -5 $O/pytest_linear3x.txt
