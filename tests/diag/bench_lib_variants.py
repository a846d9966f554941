import os, sys, subprocess, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for lib in sys.argv[1:]:
    env = dict(os.environ, NERF_HIP_LIB=os.path.join(ROOT, "nerf-pytorch_amd", "lib", lib))
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--sampling", "single", "--steps", "6", "--warmup", "2",
                          "--no-cpu-baseline"], env=env, capture_output=True, text=True)
    try:
        d = json.loads(out.stdout.strip().splitlines()[-1])
        print(lib, "ms/step %.3f" % d["ms_per_step"], {k: round(v["ms_per_step"], 3) for k, v in d["kernels"].items()}, flush=True)
    except Exception as e:
        print(lib, "FAILED", out.stderr[-300:], flush=True)
