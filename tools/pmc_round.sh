# usage (on the GPU box): bash tools/pmc_round.sh <round tag, e.g. r02a> [modes...]
# Separate rocprofv3 --pmc passes per workload and mode, counters only (no trace domains):
#   FETCH_SIZE, WRITE_SIZE                      -> <tag>_<workload>_<mode>_pmc_traffic.json (build-id stamped; bench.py's roofline.traffic)
#   SQ MFMA / LDS counters                      -> <tag>_<workload>_<mode>_pmc_mfma.txt  (k_attn_*, k_linear_*)
set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
tag=${1:-r02a}
shift || true
modes=${@:-mixed bf16}
mkdir -p gpurun_out/$tag
for w in code2 molpcba er code2-pna; do
  for m in $modes; do
    [ $w = er ] && [ $m != mixed ] && continue   # the stress workload: the headline mode only
    [ $w = code2-pna ] && [ $m != mixed ] && continue
    [ $w = molpcba ] && [ $m = fp32 ] && continue
    for c in FETCH_SIZE WRITE_SIZE; do
      rm -rf /tmp/pmc_${w}_${m}_$c
      timeout 900 rocprofv3 --pmc $c -d /tmp/pmc_${w}_${m}_$c -o res -- python bench.py --workload $w --mode $m --steps 4 --warmup 2 --no-cpu-baseline --no-kernel-timing --no-extra > gpurun_out/$tag/pmc_${w}_${m}_$c.log 2>&1 || true
    done
    f=$(find /tmp/pmc_${w}_${m}_FETCH_SIZE -name "*.db" | head -1)
    wr=$(find /tmp/pmc_${w}_${m}_WRITE_SIZE -name "*.db" | head -1)
    per=256; [ $w = code2-pna ] && per=128   # (graphs per batch of the workload's bench line)
    python tools/pmc_traffic.py $f $wr $w $m $per gpurun_out/$tag/${tag}_${w}_${m} || true
  done
done
# HBM traffic of the aggregate kernels on the stress batch (BASELINE configs[4]: the bandwidth proof; on Code2 the re-gathers hit L2)
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_aggstress_$c
  timeout 600 rocprofv3 --pmc $c -d /tmp/pmc_aggstress_$c -o res -- python tools/agg_stress.py > gpurun_out/$tag/pmc_aggstress_$c.log 2>&1 || true
done
python tools/agg_stress.py --pmc-json $(find /tmp/pmc_aggstress_FETCH_SIZE -name "*.db" | head -1) $(find /tmp/pmc_aggstress_WRITE_SIZE -name "*.db" | head -1) gpurun_out/$tag/${tag}_aggregate_stress_pmc_traffic.json || true
# MFMA / LDS issue counters of the GEMM and attention kernels (Code2 only; one SQ pass, 8 slots)
for m in $modes; do
  rm -rf /tmp/pmc_mfma_$m
  timeout 900 rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_WAVE_CYCLES SQ_INSTS_VALU -d /tmp/pmc_mfma_$m -o res -- python bench.py --workload code2 --mode $m --steps 4 --warmup 2 --no-cpu-baseline --no-kernel-timing --no-extra > gpurun_out/$tag/pmc_mfma_$m.log 2>&1 || true
  db=$(find /tmp/pmc_mfma_$m -name "*.db" | head -1)
  python tools/rocpd_pmc.py $db k_ > gpurun_out/$tag/${tag}_code2_${m}_pmc_mfma.txt 2>&1 || true
done
# the same counters on the ER stress (attention at head_dim 64 / 513 tokens, the ring GEMMs): the headline mode only
rm -rf /tmp/pmc_mfma_er
timeout 900 rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_WAVE_CYCLES SQ_INSTS_VALU -d /tmp/pmc_mfma_er -o res -- python bench.py --workload er --steps 3 --warmup 2 --no-cpu-baseline --no-kernel-timing --no-extra > gpurun_out/$tag/pmc_mfma_er.log 2>&1 || true
db=$(find /tmp/pmc_mfma_er -name "*.db" | head -1)
python tools/rocpd_pmc.py $db k_ > gpurun_out/$tag/${tag}_er_mixed_pmc_mfma.txt 2>&1 || true
ls gpurun_out/$tag
