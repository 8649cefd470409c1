"""Run bench.py's decode timing once per perf-variant build (variants/lib_*.so, tools/build_variant.sh),
one subprocess each (VTACO_HIP_LIB selects the library), interleaved over rounds."""
import glob, json, os, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
libs = sorted(glob.glob(os.path.join(root, "variants", "lib_*.so")))
names = sys.argv[1:] or [os.path.basename(l)[4:-3] for l in libs]
res = {n: [] for n in names}
for rnd in range(3):
    for n in names:
        env = dict(os.environ, VTACO_HIP_LIB=os.path.join(root, "variants", f"lib_{n}.so"))
        out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "100", "--warmup", "10",
                              "--no-cpu-baseline", "--decode-only"] + os.environ.get("BENCH_ARGS", "").split(), env=env, capture_output=True, text=True)
        try:
            j = json.loads(out.stdout.strip().splitlines()[-1])
            res[n].append(j["roofline"]["kernel_ms"])
        except Exception as e:
            res[n].append(None)
            print(n, "failed", out.stderr[-300:])
for n in names:
    print(f"{n:12s}", " ".join("%.4f" % v if v else "fail" for v in res[n]))
