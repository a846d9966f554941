"""bench_kernels.py under several library builds in ONE GPU run (diagnostic):
    python tests/diag/kernels_lib_variants.py variant_a.so variant_b.so ...   (file names under nerf-pytorch_amd/lib)"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for lib in sys.argv[1:]:
    env = dict(os.environ, NERF_HIP_LIB=os.path.join(ROOT, "nerf-pytorch_amd", "lib", lib))
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench_kernels.py")], env=env, capture_output=True, text=True)
    rows = [json.loads(l) for l in out.stdout.strip().splitlines() if l.startswith("{")]
    print(lib, {r["kernel"][:14]: round(r["us"], 1) for r in rows}, out.stderr[-200:] if not rows else "", flush=True)
