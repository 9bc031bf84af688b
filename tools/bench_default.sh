# usage (GPU box): bash tools/bench_default.sh <tag>  -- the driver's command line (python bench.py, all defaults) with its wall time
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/${1:-r04q}; mkdir -p $O
t0=$(date +%s); python bench.py --report $O/bench_default_report.json > $O/bench_default.json 2> $O/bench_default.err; t1=$(date +%s); echo "wall $((t1-t0)) s"
python - <<PY
import json
line=open("$O/bench_default.json").read().strip().splitlines()[-1]
print("line bytes", len(line)); json.loads(line)
d=json.load(open("$O/bench_default_report.json"))
print(d["metric"], d["value"], d["ms_per_step"], d["dtype"], d["scaling"])
print("roofline", d["roofline"]["kernel"], d["roofline"]["frac"], d["roofline"].get("traffic"))
print("modes", {k:(v["value"],v["ms_per_step"]) for k,v in d["modes"].items()})
print("prec", {k:(v.get("loss_rel_err"),v.get("grad_rel_l2_worst"),v.get("worst_vs_oracle_noise")) for k,v in d["precision_vs_oracle"].items() if isinstance(v,dict)})
print("cpu", d["cpu_baseline"]["value"], d["cpu_baseline"]["cores"], "host idle", d["host_enqueue_ms_per_step_idle_device"])
print("agg stress", {k:(v.get("frac") if isinstance(v,dict) else v) for k,v in d["aggregate_stress"].items() if k.startswith("gt_")})
PY
