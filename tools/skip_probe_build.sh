# usage: bash tools/skip_probe_build.sh   -- builds graphtrans_amd/libgt_skip.so: the library with every kernel launch behind
# GT_SKIP (comma-separated substrings of kernel names; a matching launch is dropped).  A what-if tool for the critical path:
#   GT_LIB_PATH=$PWD/graphtrans_amd/libgt_skip.so GT_SKIP=k_ln_bwd_d128 python bench.py --no-kernel-timing ...
# gives the step time WITHOUT that kernel family (results are garbage, timing is what is read): the upper bound of what fusing
# it away can gain under the real three-stream schedule.  Never loaded by default.
set -e
cd "$(dirname "$0")/.."
O=graphtrans_amd/csrc/build_skip; mkdir -p $O
for s in common graph_prep aggregate segment attention norm linear layers model pna embed xent optim util collate; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DGT_DEBUG_SKIP -c graphtrans_amd/csrc/$s.hip -o $O/$s.o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o graphtrans_amd/libgt_skip.so $O/*.o
echo graphtrans_amd/libgt_skip.so
