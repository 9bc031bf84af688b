cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r05r; O=gpurun_out/r05r
run() { n=$1; shift; e=$1; shift
  env $e timeout 300 python bench.py --no-cpu-baseline --no-extra --steps 100 "$@" > $O/$n.json 2>$O/$n.err
  python -c "
import json
d=json.load(open('$O/$n.json')); r=d['roofline']; k=d['kernels']
print('$n', d['value'], d['ms_per_step'], r['kernel'], r['frac'], r['avg_us'], '| dw16', k.get('k_dw16+reduce[bf16]',{}).get('avg_us'), k.get('k_dw16+reduce[bf16]',{}).get('frac'), '| lin3r_dw', k.get('k_lin3r_dw+reduce',{}).get('avg_us'), k.get('k_lin3r_dw+reduce',{}).get('frac'), k.get('k_lin3r_dw+reduce',{}).get('total_ms'), k.get('k_dw16+reduce[bf16]',{}).get('total_ms'))"
}
for i in 1 2; do
run b256 "GT_DW16_BLOCKS=256"
run b512 "GT_DW16_BLOCKS=512"
run b128 "GT_DW16_BLOCKS=128"
done
timeout 200 python tools/dw16_bench.py 2>&1 | grep -v amdgpu | tail -12
