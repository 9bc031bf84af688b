#!/usr/bin/env python
"""cProfile of the host side of bench.py's step (where does Python/launch time go?).
With --batch 8 the GPU is never the bottleneck, so the times are pure host enqueue cost."""
import cProfile, pstats, sys, os, io
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
batch = sys.argv[1] if len(sys.argv) > 1 else "8"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 100
workload = sys.argv[3] if len(sys.argv) > 3 else "code2"
sys.argv = ["bench.py", "--steps", str(steps), "--warmup", "10", "--no-cpu-baseline", "--no-kernel-timing", "--no-extra", "--batch", batch,
            "--workload", workload]
import bench
pr = cProfile.Profile()
pr.enable()
bench.main()
pr.disable()
for key, n in (("tottime", 45), ("cumulative", 70)):
    s = io.StringIO()
    pstats.Stats(pr, stream=s).sort_stats(key).print_stats(n)
    print(s.getvalue()[:12000])
