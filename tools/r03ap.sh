cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r03ap; mkdir -p $O
timeout 900 python -m pytest tests/test_hip_attention.py -x -q > $O/pytest_attn.txt 2>&1; tail -2 $O/pytest_attn.txt
cat > /tmp/ab.py <<'PY'
import sys, numpy as np, torch
sys.path.insert(0, "tools"); sys.path.insert(0, ".")
import attn_bench as ab
from graphtrans_amd import synth
b = synth.code2_like(B=256, seed=1000)
n = np.minimum(torch.bincount(b.batch).numpy(), 1000) + 1
ab.case("Code2 p=0.3", list(n), p=0.3, lpt=True)
PY
for v in new old new old; do
  case $v in old) export GT_LIB_PATH=$GRAFT_REPO_ROOT/graphtrans_amd/libgt_old_attn.so;; *) unset GT_LIB_PATH;; esac
  echo "== $v"; timeout 300 python /tmp/ab.py 2>&1 | grep kernels
  timeout 300 python bench.py --workload code2 --no-kernel-timing --no-cpu-baseline --no-extra --steps 100 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][0]); print('code2', d['value'], d['ms_per_step'], d['final_loss'])"
done 2>&1 | tee $O/ab.txt
