"""Where the HOST spends a step (cProfile around bench.py's step loop): python tools/host_profile.py [workload] [steps]"""
import cProfile, pstats, sys, io, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.argv = ["bench.py", "--workload", sys.argv[1] if len(sys.argv) > 1 else "code2", "--steps", sys.argv[2] if len(sys.argv) > 2 else "60", "--warmup", "10",
            "--no-kernel-timing", "--no-cpu-baseline", "--no-extra"]
import bench
pr = cProfile.Profile()
pr.enable()
bench.main()
pr.disable()
for key in ("tottime", "cumulative"):
    s = io.StringIO()
    pstats.Stats(pr, stream=s).sort_stats(key).print_stats(45)
    print(s.getvalue()[:9000])
# who calls the slow built-ins (one hipGetDeviceCount costs ~1 ms of host time on these boxes)
for pat in ("getDeviceCount", "is_available", "method 'to' of", "method 'contiguous'", "_lazy_init"):
    s = io.StringIO()
    pstats.Stats(pr, stream=s).sort_stats("cumulative").print_callers(pat)
    print(s.getvalue()[:3000])
