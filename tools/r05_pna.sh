# usage (GPU box): bash tools/r05_pna.sh  -- Code2-PNA: grouped GEMM tests, A/B of the 30-KB dX kernel (GT_LIN3_SMALL_LDS) and the
# weight-gradient block shape (GT_LIN3R_DW_SHAPE), timeline
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r05pna; o=gpurun_out/r05pna
timeout 900 python -m pytest tests/test_hip_linear3x.py tests/test_hip_pna.py tests/test_hip_linear3r.py -x -q > $o/tests.txt 2>&1; tail -3 $o/tests.txt
for v in "1 1" "0 1" "1 0" "1 1" "0 1"; do
  set -- $v
  GT_LIN3_SMALL_LDS=$1 GT_LIN3R_DW_SHAPE=$2 timeout 600 python bench.py --workload code2-pna --steps 100 --warmup 20 --no-cpu-baseline --no-kernel-timing --no-extra > $o/bench_s$1$2.json 2> $o/bench_s$1$2.err
  python -c "import json;d=json.loads(open('$o/bench_s$1$2.json').read().strip().splitlines()[-1]);print('small_lds=$1 dw_shape(0 = 160x160, 1 = 224x128)=$2',d['value'],d['ms_per_step'],d.get('final_loss'))"
done
bash tools/r05_timeline.sh r05pna code2-pna mixed
