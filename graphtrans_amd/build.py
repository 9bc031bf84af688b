"""Build libgraphtrans_hip.so (gfx950) in-tree with hipcc.  `python -m graphtrans_amd.build`.

hipcc cross-compiles without a GPU; the resulting .so is git-ignored but travels to the GPU box
with the repo snapshot.  Objects are rebuilt only when their source (or a header) is newer.
"""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
INCLUDE = os.path.join(os.path.dirname(HERE), "include")
LIB = os.path.join(HERE, "libgraphtrans_hip.so")
SOURCES = ["common.hip", "graph_prep.hip", "aggregate.hip", "segment.hip", "attention.hip", "norm.hip", "linear.hip", "layers.hip", "model.hip", "pna.hip", "embed.hip", "xent.hip", "optim.hip", "util.hip", "collate.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function"]


def _hipcc():
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found: cannot build libgraphtrans_hip.so")
    return exe


def build(force=False, verbose=False):
    hipcc = _hipcc()
    objdir = os.path.join(CSRC, "build")
    os.makedirs(objdir, exist_ok=True)
    headers = [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC)) if f.endswith(".h")] + [os.path.join(INCLUDE, "graphtrans_hip.h")]
    hdr_m = max(os.path.getmtime(h) for h in headers)
    objs, rebuilt = [], False
    procs = []
    for s in SOURCES:
        src = os.path.join(CSRC, s)
        if not os.path.exists(src):
            continue
        obj = os.path.join(objdir, s.replace(".hip", ".o"))
        objs.append(obj)
        if force or not os.path.exists(obj) or os.path.getmtime(obj) < max(os.path.getmtime(src), hdr_m):
            cmd = [hipcc] + FLAGS + ["-c", src, "-o", obj]
            if verbose:
                print(" ".join(cmd))
            procs.append((s, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
            rebuilt = True
    for s, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError(f"hipcc failed on {s}:\n{out.decode()}")
        if verbose and out:
            print(out.decode())
    if rebuilt or not os.path.exists(LIB) or any(os.path.getmtime(o) > os.path.getmtime(LIB) for o in objs):   # (objects compiled by hand count)
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
        r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stdout.decode()}")
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
