"""One training chunk of 8192 rays x 256 samples (2,097,152 samples, 47 GB of workspaces) against four chunks of 2048 rays: same image, same accumulated gradient (GPU box)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "nerf-pytorch_amd"))
import torch
from nerf.models import NeRF
from nerf.rendering import render_nerf
dev = torch.device("cuda", 0)
torch.manual_seed(12)
m = NeRF().to(dev)
R, S = 8192, 256
g = torch.Generator().manual_seed(1)
o = torch.randn(R, 3, generator=g); o = 4 * o / o.norm(dim=1, keepdim=True)
d = torch.nn.functional.normalize(-o + 0.3 * torch.randn(R, 3, generator=g), dim=1)
o, d = o.to(dev), d.to(dev)
tgt = torch.rand(R, 3, generator=g).to(dev)
u = torch.rand(S, generator=g).to(dev)
def grads(chunk):
    for p in m.parameters(): p.grad = None
    n = (R + chunk - 1) // chunk
    rgb = render_nerf(m, o, d, 2.0, 6.0, num_samples=S, device=dev, chunk_size=chunk, u=[u] * n)
    torch.nn.functional.mse_loss(rgb, tgt).backward()
    return rgb.detach().clone(), torch.cat([p.grad.reshape(-1) for p in m.parameters()]).clone()
r1, g1 = grads(8192)     # one chunk of 2,097,152 samples
r2, g2 = grads(2048)     # four chunks
print("rgb max diff", float((r1 - r2).abs().max()))
print("grad rel diff", float((g1 - g2).norm() / g2.norm()), "norm", float(g2.norm()))
print("mem GB", torch.cuda.max_memory_allocated() / 2**30)
