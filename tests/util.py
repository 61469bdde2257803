"""Shared helpers for the parity tests (oracle = checker, dynhor_amd = product)."""
import torch

from oracle import neus_oracle as O


def flat_from_oracle(sdf, var, col):
    """Flat fp32 parameter vector in dh_param_layout order == state_dict order."""
    return torch.cat([p.detach().reshape(-1).float() for m in (sdf, var, col)
                      for p in m.state_dict().values()]).contiguous()


def randomized_models(seed=7, device="cuda", jitter=0.05):
    """Oracle networks with geometric/default init, then jittered so weight-norm, biases and every column of
    every matrix matter (geometric init alone zeroes many columns)."""
    sdf, col, var = O.build_models(seed=seed, device=device)
    g = torch.Generator(device="cpu").manual_seed(seed + 1)
    with torch.no_grad():
        for m in (sdf, col):
            for name, p in m.named_parameters():
                noise = torch.randn(p.shape, generator=g) * jitter
                if name.endswith("weight_g"):
                    p.mul_(1.0 + noise.to(p.device))
                else:
                    p.add_(noise.to(p.device) * (0.2 if name.endswith("bias") else 1.0 / 16.0))
    return sdf, col, var
