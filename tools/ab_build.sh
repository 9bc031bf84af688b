# usage (here, no GPU): bash tools/ab_build.sh <git rev> [file.hip ...]   -- builds graphtrans_amd/libgt_old.so: the in-tree objects with the
# named csrc/*.hip files (default: every file that differs) taken from <git rev>; tools/ab.sh then alternates old / new on one GPU box.
# (The library must be built first: python -m graphtrans_amd.build.  libgt_old.so is git-ignored and travels with the snapshot.)
set -e
cd "$(dirname "$0")/.."
REV=${1:?git revision}; shift || true
FILES=${@:-$(git diff --name-only $REV -- graphtrans_amd/csrc | grep '\.hip$' | xargs -n1 basename)}
HDRS=$(git diff --name-only $REV -- graphtrans_amd/csrc include | grep '\.h$' || true)
[ -n "$HDRS" ] && echo "note: headers differ from $REV ($HDRS): only the named .hip files are rebuilt against the CURRENT headers"
OBJS=$(ls graphtrans_amd/csrc/build/*.o)
for f in $FILES; do
  git show $REV:graphtrans_amd/csrc/$f > graphtrans_amd/csrc/_old_$f
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function -c graphtrans_amd/csrc/_old_$f -o /tmp/_old_${f%.hip}.o
  rm graphtrans_amd/csrc/_old_$f
  OBJS=$(echo "$OBJS" | grep -v "/${f%.hip}.o") ; OBJS="$OBJS /tmp/_old_${f%.hip}.o"
done
hipcc --offload-arch=gfx950 -shared -fPIC -o graphtrans_amd/libgt_old.so $OBJS
ls -la graphtrans_amd/libgt_old.so
