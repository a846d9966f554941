"""Forward workspace of two library builds on the same inputs, compared region by region (diagnostic):
    python tests/diag/mask_diff.py dump out.npy      (with NERF_HIP_LIB set)   |   python tests/diag/mask_diff.py cmp a.npy b.npy"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "nerf-pytorch_amd")); sys.path.insert(0, ROOT)
if sys.argv[1] == "dump":
    import torch
    from nerf import _hip
    from nerf.models import NeRF
    from oracle import nerf_oracle as O
    m = NeRF(); m.load_state_dict(O.closed_form_params(O.nerf_param_shapes())); m = m.cuda()
    g = torch.Generator().manual_seed(0)
    M = 256
    pts = (torch.rand(M, 3, generator=g) * 2 - 1).cuda(); dirs = torch.nn.functional.normalize(torch.randn(M, 3, generator=g), dim=1).cuda()
    colors, density, ws = _hip.mlp_fwd(m.model_id, m.packed_weights(True), M, train=True, points=pts, dirs=dirs)
    torch.cuda.synchronize()
    np.save(sys.argv[2], ws.view(torch.int32).cpu().numpy())
else:
    a, b = np.load(sys.argv[2]), np.load(sys.argv[3])
    Mp = 256
    per = a.size // Mp
    print("floats per sample", per, "equal", np.array_equal(a, b))
    A, B = a.reshape(per, Mp), b.reshape(per, Mp)
    bad = [i for i in range(per) if not np.array_equal(A[i], B[i])]
    print("rows differing:", bad[:40], len(bad))
    for i in bad[:4]:
        d = np.nonzero(A[i] != B[i])[0]
        print(i, d[:8], [hex(int(x) & 0xffffffff) for x in A[i][d[:4]]], [hex(int(x) & 0xffffffff) for x in B[i][d[:4]]])
