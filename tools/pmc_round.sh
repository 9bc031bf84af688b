# usage (on the GPU box): bash tools/pmc_round.sh <round tag, e.g. r01p>
# Two separate rocprofv3 --pmc passes per workload (FETCH_SIZE, WRITE_SIZE), counters only: no trace domains.
set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
tag=${1:-r01p}
mkdir -p gpurun_out/$tag
for w in code2 molpcba; do
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/pmc_${w}_$c
    timeout 900 rocprofv3 --pmc $c -d /tmp/pmc_${w}_$c -o res -- python bench.py --workload $w --steps 4 --warmup 2 --no-cpu-baseline --no-kernel-timing > gpurun_out/$tag/pmc_${w}_$c.log 2>&1 || true
  done
  f=$(find /tmp/pmc_${w}_FETCH_SIZE -name "*.db" | head -1)
  wr=$(find /tmp/pmc_${w}_WRITE_SIZE -name "*.db" | head -1)
  python tools/pmc_traffic.py $f $wr $w bf16 256 gpurun_out/$tag/${tag}_${w} || true
done
ls gpurun_out/$tag
