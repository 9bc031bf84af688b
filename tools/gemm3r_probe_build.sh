# usage: bash tools/gemm3r_probe_build.sh [masks...]   -- builds tools/gemm3_probe_r<mask> (git-ignored; they travel to the GPU box)
cd "$(dirname "$0")/.."
for m in ${@:-0 1 2 3 4 7 8}; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -DW3R_ABL=$m $W3R_EXTRA -I graphtrans_amd/csrc -I include -o tools/gemm3_probe_r$m$W3R_SUFFIX tools/gemm3r_probe.hip &
done
wait
ls tools/gemm3_probe_r*
