# usage (GPU box): bash tools/r05_pna.sh  -- grouped bf16x6 tower GEMMs: tests, then the Code2-PNA line with and without
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r05pna; o=gpurun_out/r05pna
./tools/gemm3_probe_dwpna0 | tee $o/dw_pna_probe_xcd.txt
PYTHONPATH=. python tools/pna_gemm_bench.py 2>&1 | grep -v amdgpu.ids | tee $o/pna_gemm_bench.txt
timeout 900 python -m pytest tests/test_hip_linear3x.py tests/test_hip_pna.py tests/test_hip_linear.py tests/test_hip_linear3r.py -x -q > $o/tests.txt 2>&1; tail -3 $o/tests.txt
for v in "1 1" "1 0" "0 0"; do
  set -- $v
  GT_PNA_TOWER_IMAGES=$1 GT_LIN3_GROUPED_DW=$2 timeout 600 python bench.py --workload code2-pna --steps 100 --warmup 20 --no-cpu-baseline --no-kernel-timing --no-extra > $o/bench_img$1$2.json 2> $o/bench_img$1$2.err
  python -c "import json;d=json.loads(open('$o/bench_img$1$2.json').read().strip().splitlines()[-1]);print('images=$1 dw=$2',d['value'],d['ms_per_step'],d.get('final_loss'))"
done
bash tools/r05_timeline.sh r05pna code2-pna mixed
