"""Generate tests/golden/neus_small.npz from the oracle (fp32, CPU).  The reference holds no implementation or
vectors for this path, so this fixture pins the oracle against regressions of itself: cfg1-shaped
(128x128 frame geometry, 48 rays x (32+32) samples)."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import neus_oracle as O  # noqa: E402


def main():
    torch.set_num_threads(1)
    seed = 1234
    sdf, col, var = O.build_models(seed=seed)
    r = O.NeuSRenderer(None, sdf, var, col, 32, 32, 0, 4, 1.0)
    g = torch.Generator().manual_seed(99)
    B = 48
    o = torch.nn.functional.normalize(torch.randn(B, 3, generator=g), dim=-1) * 2.2
    d = torch.nn.functional.normalize((torch.rand(B, 3, generator=g) - 0.5) * 0.6 - o, dim=-1)
    rays = torch.cat([o, d, torch.rand(B, 3, generator=g), (torch.rand(B, 1, generator=g) > 0.4).float(),
                      (torch.rand(B, 1, generator=g) > 0.2).float(),
                      torch.nn.functional.normalize(torch.randn(B, 3, generator=g), dim=-1)], -1)
    t_rand = torch.rand(B, 1, generator=g)
    R = torch.linalg.qr(torch.randn(3, 3, generator=g))[0]
    near, far = O.near_far_from_sphere(o, d)
    out = r.render(o, d, near, far, cos_anneal_ratio=0.25, t_rand=t_rand)
    losses = O.neus_losses(out, rays[:, 6:9], rays[:, 9:10], rays[:, 10:11], 0.1, 0.1, 0.05, rays[:, 11:14], R)
    losses["loss"].backward()
    gn = float(np.sqrt(sum(float((p.grad ** 2).sum()) for m in (sdf, var, col) for p in m.parameters())))
    dst = os.path.join(os.path.dirname(os.path.abspath(__file__)), "neus_small.npz")
    np.savez(dst, seed=seed, rays=rays.numpy(), t_rand=t_rand.numpy(), R=R.numpy(), cos_anneal=0.25,
             z_vals=out["z_vals"].detach().numpy(), color_fine=out["color_fine"].detach().numpy(),
             weight_sum=out["weight_sum"].detach().numpy(), loss=float(losses["loss"]), grad_norm=gn)
    print("wrote", dst, os.path.getsize(dst), "bytes; loss", float(losses["loss"]), "grad norm", gn)


if __name__ == "__main__":
    main()
