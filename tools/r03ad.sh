cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r03ad; mkdir -p $O
rm -rf /tmp/pd; rocprofv3 --kernel-trace --stats -d /tmp/pd -o res -- python tools/dw16_bench.py > $O/log.txt 2>&1
db=$(find /tmp/pd -name "*.db" | head -1)
python - "$db" <<'PY'
import sqlite3,sys
cur=sqlite3.connect(sys.argv[1]).cursor()
rows=list(cur.execute("select name, grid_x*grid_y*grid_z/(workgroup_x*workgroup_y*workgroup_z), end-start from kernels order by start"))
import collections
d=collections.defaultdict(list)
for n,g,t in rows:
    if 'k_dw16' in n or 'split_reduce' in n: d[(n[:60],g)].append(t/1e3)
for k,v in d.items():
    v=sorted(v); print(k, len(v), 'median %.1f us'%v[len(v)//2], 'min %.1f'%v[0])
PY
