"""A/B timing of library builds inside ONE GPU run (box-to-box variation is about 2 %):
    python tests/diag/bench_lib_variants.py [--sampling single|hier] [--mode train|infer] [--model nerf|siren] [--train-precision bf16x3] libnerf_hip.so variant_x.so ...
Each library (a file name under nerf-pytorch_amd/lib) runs bench.py in a child process, the list is walked twice."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
args = sys.argv[1:]
sampling, mode, model = "hier", "train", "nerf"
extra = []
while args and args[0].startswith("--"):
    if args[0] == "--sampling":
        sampling = args[1]
    elif args[0] == "--mode":
        mode = args[1]
    elif args[0] == "--model":
        model = args[1]
    elif args[0] == "--train-precision":
        extra += ["--train-precision", args[1]]
    args = args[2:]
for rep in range(2):
    for lib in args:
        env = dict(os.environ, NERF_HIP_LIB=os.path.join(ROOT, "nerf-pytorch_amd", "lib", lib))
        out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--sampling", sampling, "--mode", mode, "--model", model, "--steps", "8",
                              "--warmup", "3", "--no-cpu-baseline"] + extra, env=env, capture_output=True, text=True)
        try:
            d = json.loads(out.stdout.strip().splitlines()[-1])
            print(lib, "ms/step %.3f" % d["ms_per_step"], {k: round(v["ms_per_step"], 3) for k, v in d["kernels"].items()}, flush=True)
        except Exception:
            print(lib, "FAILED", out.stderr[-300:], flush=True)
