cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r03ah; mkdir -p $O
cat > /tmp/ab.py <<'PY'
import sys, numpy as np, torch
sys.path.insert(0, "tools"); sys.path.insert(0, ".")
import attn_bench as ab
from graphtrans_amd import synth
b = synth.code2_like(B=256, seed=1000)
n = np.minimum(torch.bincount(b.batch).numpy(), 1000) + 1
ab.case("Code2 p=0.3", list(n), p=0.3, lpt=True)
ab.case("Code2 p=0", list(n), p=0.0, lpt=True)
PY
for v in new old; do
  case $v in old) export GT_LIB_PATH=$GRAFT_REPO_ROOT/graphtrans_amd/libgt_old_attn.so;; noshare) export GT_LIB_PATH=$GRAFT_REPO_ROOT/graphtrans_amd/libgt_noshare.so;; *) unset GT_LIB_PATH;; esac
  rm -rf /tmp/pa; rocprofv3 --kernel-trace -d /tmp/pa -o res -- python /tmp/ab.py > $O/log_$v.txt 2>&1
  db=$(find /tmp/pa -name "*.db" | head -1)
  echo "== $v"; grep kernels $O/log_$v.txt
  python - "$db" <<'PY'
import sqlite3,sys,collections
cur=sqlite3.connect(sys.argv[1]).cursor()
d=collections.defaultdict(list)
for n,t in cur.execute("select name, end-start from kernels order by start"):
    if 'k_attn' in n: d[n.split('<')[0].split('::')[-1]].append(t/1e3)
for k,v in d.items():
    h=len(v)//2; a=sorted(v[:h]); b=sorted(v[h:]); print(k, 'p=0.3 median %.1f'%a[len(a)//2], ' p=0 median %.1f'%b[len(b)//2])
PY
done 2>&1 | tee $O/ab.txt
