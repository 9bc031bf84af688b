cd $GRAFT_REPO_ROOT
O=gpurun_out/r03e; mkdir -p $O
for v in 0 3 19 35 51 67 7 71 87 119 127; do timeout 120 ./tools/gemm3_probe_$v | head -1; done > $O/probe.txt 2>&1
cat $O/probe.txt
