# usage (GPU box): bash tools/ab.sh [bench args...]   -- A/B of graphtrans_amd/libgt_old.so (previous build) vs the current library
for v in old new old new; do
  unset GT_LIB_PATH; [ $v = old ] && export GT_LIB_PATH=$PWD/graphtrans_amd/libgt_old.so
  python bench.py --steps 80 --warmup 10 --no-kernel-timing --no-cpu-baseline "$@" 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', d['value'], d['ms_per_step'])"
done
