# ER (config 5) round-6 baseline: kernel summary + per-stream timeline of HEAD, the weight-gradient and GEMM micro-benches at the ER shapes
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r06er; mkdir -p $O
python tools/dw16_bench.py 131328 > $O/dw16_bench_er.txt 2>&1; cat $O/dw16_bench_er.txt
python tools/gemm1_bench.py 131328 > $O/gemm1_bench_er.txt 2>&1; tail -20 $O/gemm1_bench_er.txt
python bench.py --workload er --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timing --no-extra > $O/bench_er_clean.json 2>/dev/null; cut -c1-300 $O/bench_er_clean.json
bash tools/prof_one.sh r06er er --steps 20 --warmup 5 | head -50
db=$(find /tmp/prof_er -name "*.db" | head -1)
python tools/timeline_json.py $db $O/r06er_timeline_er.json 12 > $O/r06er_timeline_er.txt 2>&1; head -40 $O/r06er_timeline_er.txt
