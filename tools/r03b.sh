set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r03b; mkdir -p $O
timeout 600 python -m pytest tests/test_hip_engine.py -x -q > $O/pytest_engine.txt 2>&1
for w in code2 molpcba; do
  rm -rf /tmp/prof_$w
  rocprofv3 --kernel-trace --stats -d /tmp/prof_$w -o res -- python bench.py --workload $w --steps 30 --warmup 10 --no-cpu-baseline --no-kernel-timing --no-extra > $O/prof_$w.log 2>&1
  db=$(find /tmp/prof_$w -name "*.db" | head -1)
  python tools/timeline.py $db $O/timeline_$w.txt 3 > /dev/null 2>&1
  python tools/rocpd_summary.py $db 40 $O/r03b_${w}_mixed > /dev/null 2>&1
done
rm -rf /tmp/prof_hip
rocprofv3 --hip-trace --stats -d /tmp/prof_hip -o res -- python bench.py --batch 8 --steps 100 --warmup 10 --no-cpu-baseline --no-kernel-timing --no-extra > $O/prof_hip.log 2>&1
find /tmp/prof_hip -type f | head -20 > $O/prof_hip_files.txt
db=$(find /tmp/prof_hip -name "*.db" | head -1)
python - "$db" > $O/hip_api_stats.txt 2>&1 <<'PY'
import sqlite3, sys
cur = sqlite3.connect(sys.argv[1]).cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")]
print(tabs)
for t in tabs:
    if 'top' in t or 'stat' in t.lower():
        try:
            rows = list(cur.execute(f"select * from {t} limit 60"))
            print('==', t, [d[0] for d in cur.description])
            for r in rows: print(r)
        except Exception as e:
            print(t, e)
PY
