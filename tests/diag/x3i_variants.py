"""Three-term inference forward: the 16-sample-tile variants of mlp_x3i.hip next to the fp32 kernel (development A/B)."""
import ctypes, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "nerf-pytorch_amd"))
import torch
from nerf import _hip
from nerf.models import NeRF, Siren

dev = "cuda"
torch.manual_seed(0)
SIREN = len(sys.argv) > 1 and sys.argv[1] == "siren"
m = (Siren() if SIREN else NeRF()).to(dev)
MODEL = _hip.MODEL_SIREN if SIREN else _hip.MODEL_NERF
M = int(os.environ.get("M", 1 << 20))
x = torch.rand(M, 3, device=dev) * 2 - 1
d = torch.nn.functional.normalize(torch.randn(M, 3, device=dev), dim=1)
params = [p.detach().contiguous() for p in m.parameters()]
img32 = _hip.mlp_pack(MODEL, params)
FLOP = 2 * (561920 if SIREN else 658944)
lib = _hip.lib
P = ctypes.c_void_p
lib.nerf_x3i_bytes.restype = ctypes.c_size_t
lib.nerf_x3i_bytes.argtypes = [ctypes.c_int, ctypes.c_int]
lib.nerf_x3i_pack.restype = ctypes.c_int
lib.nerf_x3i_pack.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.POINTER(P), P, P]
lib.nerf_x3i_fwd.restype = ctypes.c_int
lib.nerf_x3i_fwd.argtypes = [ctypes.c_int, ctypes.c_int, P, ctypes.c_int64, P, P, P, P, P, ctypes.c_int, ctypes.c_int,
                             ctypes.c_double, ctypes.c_double, P, P, P]


def stream():
    return torch.cuda.current_stream().cuda_stream


def pack(variant):
    n = lib.nerf_x3i_bytes(MODEL, variant)
    img = torch.empty(n, dtype=torch.uint8, device=dev)
    arr = (P * len(params))(*[p.data_ptr() for p in params])
    rc = lib.nerf_x3i_pack(MODEL, variant, arr, img.data_ptr(), stream())
    assert rc == 0, _hip.lib.nerf_last_error()
    return img


def fwd(variant, img, colors=True):
    c = torch.empty(M, 3, device=dev) if colors else None
    s = torch.empty(M, device=dev)
    rc = lib.nerf_x3i_fwd(MODEL, variant, img.data_ptr(), M, x.data_ptr(), d.data_ptr(), None, None, None, 0, 0, 0.0, 1.0,
                          c.data_ptr() if colors else None, s.data_ptr(), stream())
    assert rc == 0, _hip.lib.nerf_last_error()
    return c, s


def run(name, f, reps=5):
    out = f()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        f()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / reps * 1e3
    print(f"{name:14s} {ms:8.3f} ms  {M * FLOP / ms / 1e9:7.1f} TFLOP/s (fp32-equivalent)", flush=True)
    return out


c32, s32, _ = run("fp32", lambda: _hip.mlp_fwd(MODEL, img32, M, points=x, dirs=d, train=False))
img3 = _hip.mlp_pack_bf16x3(MODEL, params)
c3, s3, _ = run("bf16x3 (32)", lambda: _hip.mlp_fwd_bf16(MODEL, img3, M, points=x, dirs=d))
outs = {"bf16x3 (32)": (c3, s3)}
variants = [int(v) for v in os.environ.get("VARIANTS", "0,1,2").split(",")]
for v in variants:
    img = pack(v)
    outs[f"x3i v{v}"] = run(f"x3i v{v}", lambda: fwd(v, img))
    _, sd = run(f"x3i v{v} dens", lambda: fwd(v, img, colors=False))
    print(f"   density-only equals full: {bool(torch.equal(sd, outs[f'x3i v{v}'][1]))}")
for rep in range(2):   # second pass: the order of the runs does not matter
    for v in variants:
        img = pack(v)
        run(f"x3i v{v} again", lambda: fwd(v, img))
for name, (c, s) in outs.items():
    print(f"{name:14s} colors rel {float((c - c32).norm() / c32.norm()):.3e} max abs {float((c - c32).abs().max()):.3e}; "
          f"density rel {float((s - s32).norm() / (s32.norm() + 1e-30)):.3e} max abs {float((s - s32).abs().max()):.3e}")
