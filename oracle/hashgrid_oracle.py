"""Plain-PyTorch restatement of the instant-nsr-pl style geometry/texture networks named by BASELINE.json configs[3]
(multires hash-grid encoding + shallow MLPs) -- TEST INFRASTRUCTURE, see oracle/__init__.py.

PARITY UNPINNED, doubly so: the reference only *mentions* this variant (README.md:11,13: "Replace NeuS with
instant-nsr-pl", code on an unmounted `dev` branch); nothing of it is in /root/reference, and neither
`bennyguo/instant-nsr-pl` nor `NVlabs/tiny-cuda-nn` is on disk.  This file restates, from the published descriptions
(Mueller et al. 2022 "Instant NGP" §3 for the encoding; instant-nsr-pl's neus config for the network shapes), the
pieces SURVEY.md §8f n3 lists:
  * HashGridEncoding: L=16 levels x F=2 features, table size T=2^19 per level, base resolution 16, geometric growth to
    2048; per level: dense index when the level's grid fits the table, else the spatial hash
    (x*1) ^ (y*2654435761) ^ (z*805459861) mod T; trilinear interpolation of the 8 corner features.
  * HashSDFNetwork: [2x-1 (3), enc(32)] -> Linear(35,64) -> Softplus(beta=100) -> Linear(64,13), weight-normed, sphere
    initialised; sdf = out[:,0], feature = out (13); normals by central finite differences (eps 1e-3).
  * SHRenderingNetwork: [feature(13), SH_4(dir)(16), normal(3)] -> 64 -> 64 -> 3, ReLU, sigmoid.
The NeuS sampler / compositing / losses are the ones of oracle/neus_oracle.py (this variant keeps hierarchical
up-sampling instead of instant-nsr-pl's occupancy-grid marching -- a stated build decision).
"""
from __future__ import annotations

import math

import torch
import torch.nn as nn
import torch.nn.functional as F

PRIMES = (1, 2654435761, 805459861)


class HashGridEncoding(nn.Module):
    MAX_ROWS_PER_GATHER = 1 << 18           # points per index_select (x 128 corner rows x 2 features: 256 MB in fp32, 512 MB in fp64)

    def __init__(self, n_levels=16, n_features=2, log2_hashmap_size=19, base_resolution=16, max_resolution=2048):
        super().__init__()
        self.L, self.F, self.T = n_levels, n_features, 1 << log2_hashmap_size
        self.per_level_scale = math.exp((math.log(max_resolution) - math.log(base_resolution)) / (n_levels - 1))
        self.scales, self.resolutions, self.offsets, self.sizes, self.dense = [], [], [], [], []
        off = 0
        for l in range(n_levels):
            scale = base_resolution * self.per_level_scale ** l - 1.0
            res = int(math.ceil(scale)) + 1
            n = res ** 3
            dense = n <= self.T
            size = ((n + 7) // 8) * 8 if dense else self.T
            self.scales.append(scale); self.resolutions.append(res); self.offsets.append(off)
            self.sizes.append(size); self.dense.append(dense)
            off += size
        self.n_entries = off
        self.table = nn.Parameter(torch.empty(off, n_features).uniform_(-1e-4, 1e-4))
        self.n_output_dims = n_levels * n_features

    def level_index(self, l: int, ix, iy, iz):
        """Table row of integer grid corner (ix,iy,iz) at level l (int64 tensors)."""
        res = self.resolutions[l]
        m = 0xFFFFFFFF
        # grid coordinates are uint32 (tcnn): points outside the box (the renderer's last mid-point can overshoot the unit
        # sphere) wrap modulo 2^32 before the level's own modulo -- defined, harmless, and what the HIP kernels do
        ix, iy, iz = ix & m, iy & m, iz & m
        if self.dense[l]:
            # corner coordinates reach `res` at the far faces (pos = x*scale + 0.5): wrap inside the level like tcnn does
            idx = ((ix + iy * res + iz * (res * res)) & m) % self.sizes[l]
        else:
            idx = ((ix * PRIMES[0]) & m) ^ ((iy * PRIMES[1]) & m) ^ ((iz * PRIMES[2]) & m)
            idx = idx % self.T
        return idx + self.offsets[l]

    def forward(self, x01: torch.Tensor) -> torch.Tensor:
        """x01 [N,3] in [0,1] -> [N, L*F].  The arithmetic of forward_loop below (per level: trilinear blend of the 8 corner rows) with
        ALL corner rows of all levels fetched by ONE index_select: in eager PyTorch every `table[idx]` of the loop form allocates and
        zero-fills a table-sized (98 MB) gradient in its backward, 128 times per evaluation and 7 evaluations per sample -- 0.45 s per
        2048-ray training iteration on a MI355X, which priced a paired PSNR seed of the hash family at 15 GPU-minutes (round 6)."""
        n = x01.shape[0]
        if n > self.MAX_ROWS_PER_GATHER:
            # chunked: the gathered rows of one call stay below 2^31 bytes -- round 6: at 1.5 M points in fp64 (3.2 GB of rows) the GPU
            # index_select returned rows of the wrong entries (the fp32 call of the same size, 1.6 GB, and every CPU call are correct)
            return torch.cat([self.forward(x01[i:i + self.MAX_ROWS_PER_GATHER]) for i in range(0, n, self.MAX_ROWS_PER_GATHER)], dim=0)
        idxs, ws = [], []
        for l in range(self.L):
            pos = x01 * self.scales[l] + 0.5
            pg = torch.floor(pos)
            w = pos - pg
            pg = pg.to(torch.int64)
            for dz in (0, 1):
                for dy in (0, 1):
                    for dx in (0, 1):
                        idxs.append(self.level_index(l, pg[:, 0] + dx, pg[:, 1] + dy, pg[:, 2] + dz))
                        wx = w[:, 0] if dx else 1 - w[:, 0]
                        wy = w[:, 1] if dy else 1 - w[:, 1]
                        wz = w[:, 2] if dz else 1 - w[:, 2]
                        ws.append(wx * wy * wz)
        idx = torch.stack(idxs, dim=1)                                     # [N, L*8]
        wgt = torch.stack(ws, dim=1).view(n, self.L, 8, 1)
        rows = self.table.index_select(0, idx.reshape(-1)).view(n, self.L, 8, self.F).to(x01.dtype)
        prod = wgt * rows
        acc = prod[:, :, 0]
        for c in range(1, 8):                                              # the loop form's summation order
            acc = acc + prod[:, :, c]
        return acc.reshape(n, self.L * self.F)

    def forward_loop(self, x01: torch.Tensor) -> torch.Tensor:
        """The same encoding level by level, corner by corner (rounds 1-5's form; tests/test_cpu_oracle_hashgrid.py holds the two equal)."""
        outs = []
        for l in range(self.L):
            pos = x01 * self.scales[l] + 0.5
            pg = torch.floor(pos)
            w = pos - pg
            pg = pg.to(torch.int64)
            acc = 0
            for dz in (0, 1):
                for dy in (0, 1):
                    for dx in (0, 1):
                        idx = self.level_index(l, pg[:, 0] + dx, pg[:, 1] + dy, pg[:, 2] + dz)
                        wx = w[:, 0] if dx else 1 - w[:, 0]
                        wy = w[:, 1] if dy else 1 - w[:, 1]
                        wz = w[:, 2] if dz else 1 - w[:, 2]
                        acc = acc + (wx * wy * wz)[:, None] * self.table[idx].to(x01.dtype)
            outs.append(acc)
        return torch.cat(outs, dim=-1)


class _WNLinear(nn.Module):
    def __init__(self, weight, bias):
        super().__init__()
        self.bias = nn.Parameter(bias.clone())
        self.weight_g = nn.Parameter(weight.norm(dim=1, keepdim=True).clone())
        self.weight_v = nn.Parameter(weight.clone())

    def forward(self, x):
        return F.linear(x, self.weight_g * self.weight_v / self.weight_v.norm(dim=1, keepdim=True), self.bias)


class HashSDFNetwork(nn.Module):
    """instant-nsr-pl VolumeSDF shape: hash grid + 1 hidden layer of 64, sphere init, feature_dim 13."""

    def __init__(self, radius=1.0, n_hidden=64, feature_dim=13, sphere_init_radius=0.5, fd_eps=1e-3, **enc_kwargs):
        super().__init__()
        self.radius, self.fd_eps, self.feature_dim = radius, fd_eps, feature_dim
        self.encoding = HashGridEncoding(**enc_kwargs)
        d_in = 3 + self.encoding.n_output_dims
        w0 = torch.zeros(n_hidden, d_in); b0 = torch.zeros(n_hidden)
        nn.init.normal_(w0[:, :3], 0.0, math.sqrt(2) / math.sqrt(n_hidden))          # sphere init: xyz only
        w1 = torch.empty(feature_dim, n_hidden); b1 = torch.empty(feature_dim)
        nn.init.normal_(w1, mean=math.sqrt(math.pi) / math.sqrt(n_hidden), std=1e-4)
        nn.init.constant_(b1, -sphere_init_radius)
        self.lin0 = _WNLinear(w0, b0)
        self.lin1 = _WNLinear(w1, b1)

    def forward(self, x):
        x01 = (x + self.radius) / (2 * self.radius)
        inp = torch.cat([x01 * 2 - 1, self.encoding(x01)], dim=-1)
        h = F.softplus(self.lin0(inp), beta=100)
        out = self.lin1(h)
        # NeuSRenderer convention: column 0 = sdf, the rest = feature vector; instant-nsr-pl's feature is ALL 13 outputs
        return torch.cat([out[:, :1], out], dim=-1)

    def sdf(self, x):
        return self.forward(x)[:, :1]

    def gradient(self, x):
        """Central finite differences (instant-nsr-pl grad_type 'finite_difference'): [N,1,3], differentiable wrt params."""
        eps = self.fd_eps
        offs = torch.eye(3, device=x.device, dtype=x.dtype) * eps
        n = x.shape[0]
        # the six shifted evaluations as ONE batch (same values as six calls; one encoding gather instead of six)
        s = self.sdf(torch.cat([x + offs[0], x - offs[0], x + offs[1], x - offs[1], x + offs[2], x - offs[2]], dim=0)).view(6, n, 1)
        g = [(s[2 * i] - s[2 * i + 1]) * (0.5 / eps) for i in range(3)]
        return torch.cat(g, dim=-1).unsqueeze(1)


def sh4(d: torch.Tensor) -> torch.Tensor:
    """Real spherical harmonics up to degree 4 (16 coefficients) of unit directions d [N,3]."""
    x, y, z = d[:, 0], d[:, 1], d[:, 2]
    xx, yy, zz, xy, yz, xz = x * x, y * y, z * z, x * y, y * z, x * z
    return torch.stack([
        torch.full_like(x, 0.28209479177387814),
        -0.48860251190291987 * y, 0.48860251190291987 * z, -0.48860251190291987 * x,
        1.0925484305920792 * xy, -1.0925484305920792 * yz, 0.94617469575755997 * zz - 0.31539156525251999,
        -1.0925484305920792 * xz, 0.54627421529603959 * (xx - yy),
        0.59004358992664352 * y * (-3.0 * xx + yy), 2.8906114426405538 * xy * z,
        0.45704579946446572 * y * (1.0 - 5.0 * zz), 0.3731763325901154 * z * (5.0 * zz - 3.0),
        0.45704579946446572 * x * (1.0 - 5.0 * zz), 1.4453057213202769 * z * (xx - yy),
        0.59004358992664352 * x * (-xx + 3.0 * yy)], dim=-1)


class SHRenderingNetwork(nn.Module):
    """instant-nsr-pl VolumeRadiance shape: [feature(13), SH4(dir)(16), normal(3)] -> 64 -> 64 -> 3, sigmoid."""

    def __init__(self, feature_dim=13, n_hidden=64):
        super().__init__()
        dims = [feature_dim + 16 + 3, n_hidden, n_hidden, 3]
        for l in range(3):
            lin = nn.Linear(dims[l], dims[l + 1])
            setattr(self, "lin" + str(l), _WNLinear(lin.weight.data, lin.bias.data))

    def forward(self, points, normals, view_dirs, feature_vectors):
        x = torch.cat([feature_vectors, sh4(view_dirs), normals], dim=-1)
        x = F.relu(self.lin0(x))
        x = F.relu(self.lin1(x))
        return torch.sigmoid(self.lin2(x))


def build_models(seed=1234, device="cpu", dtype=torch.float32, **enc_kwargs):
    with torch.random.fork_rng(devices=[]):
        torch.manual_seed(seed)
        sdf = HashSDFNetwork(**enc_kwargs)
        col = SHRenderingNetwork()
    return sdf.to(device=device, dtype=dtype), col.to(device=device, dtype=dtype)
