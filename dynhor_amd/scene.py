"""Synthetic hand-held-object sequence generator (SURVEY.md §8d "Synthetic inputs"): an analytic SDF object inside
the radius-0.5 ball (reference mesh normalisation, ObjTracker/run.py:110-112), F cameras on a seeded orbit with the
reference intrinsics (f = 1.2*min(H,W), c = (W//2, H//2): run.py:119-123), poses stored as x_cam = R x_obj + T
(run.py:166), SAM-style label maps 1 object / 0 background / -1 hand (run.py:66) from seeded ellipse "hand" blobs,
and camera-frame monocular normals as u8.  Data generation only -- plain torch ops, any device."""
from __future__ import annotations

import math

import torch


def scene_sdf(p: torch.Tensor) -> torch.Tensor:
    """sphere(r=0.30 at (0.06,0,0)) U rounded box (half extents 0.16,0.11,0.20, radius 0.04 at (-0.10,0.02,0)), smooth union."""
    c1 = torch.tensor([0.06, 0.0, 0.0], device=p.device, dtype=p.dtype)
    d1 = torch.linalg.norm(p - c1, dim=-1) - 0.30
    c2 = torch.tensor([-0.10, 0.02, 0.0], device=p.device, dtype=p.dtype)
    he = torch.tensor([0.16, 0.11, 0.20], device=p.device, dtype=p.dtype)
    q = (p - c2).abs() - he
    d2 = torch.linalg.norm(q.clamp(min=0.0), dim=-1) + q.max(dim=-1).values.clamp(max=0.0) - 0.04
    k = 0.05
    h = (0.5 + 0.5 * (d2 - d1) / k).clamp(0.0, 1.0)
    return d2 * (1 - h) + d1 * h - k * h * (1 - h)


def _normal(p):
    eps = 1e-4
    e = torch.eye(3, device=p.device, dtype=p.dtype) * eps
    g = torch.stack([scene_sdf(p + e[i]) - scene_sdf(p - e[i]) for i in range(3)], dim=-1)
    return g / (torch.linalg.norm(g, dim=-1, keepdim=True) + 1e-12)


def look_at_pose(cam_pos: torch.Tensor, up=None):
    """OpenCV camera (x right, y down, z forward) looking at the origin; returns R (object->camera rows), T."""
    up = torch.tensor([0.0, 0.0, 1.0], dtype=cam_pos.dtype) if up is None else up
    zf = -cam_pos / torch.linalg.norm(cam_pos)
    xr = torch.linalg.cross(zf, up)
    xr = xr / torch.linalg.norm(xr)
    yd = torch.linalg.cross(zf, xr)
    R = torch.stack([xr, yd, zf], dim=0)
    T = -(R @ cam_pos)
    return R, T


@torch.no_grad()
def make_sequence(n_frames=64, H=512, W=512, seed=4321, device="cpu", hand=True, chunk=1 << 18):
    """Returns dict(rgb u8 [F,H,W,3], label i8 [F,H,W], normal u8 [F,H,W,3], R [F,3,3], T [F,3], K [3,3]) on device."""
    g = torch.Generator(device="cpu").manual_seed(seed)
    dev = torch.device(device)
    f = 1.2 * min(H, W)
    K = torch.tensor([[f, 0, W // 2], [0, f, H // 2], [0, 0, 1]], dtype=torch.float32)
    Rs, Ts = [], []
    for i in range(n_frames):
        az = 2 * math.pi * (i + torch.rand(1, generator=g).item() * 0.5) / n_frames
        el = (torch.rand(1, generator=g).item() - 0.5) * 1.2
        rad = 2.0 + 0.5 * torch.rand(1, generator=g).item()
        pos = torch.tensor([rad * math.cos(el) * math.cos(az), rad * math.cos(el) * math.sin(az), rad * math.sin(el)])
        R, T = look_at_pose(pos)
        Rs.append(R); Ts.append(T)
    R = torch.stack(Rs).float(); T = torch.stack(Ts).float()
    rgb = torch.zeros(n_frames, H, W, 3, dtype=torch.uint8, device=dev)
    label = torch.zeros(n_frames, H, W, dtype=torch.int8, device=dev)
    normal = torch.full((n_frames, H, W, 3), 128, dtype=torch.uint8, device=dev)
    ys, xs = torch.meshgrid(torch.arange(H, device=dev), torch.arange(W, device=dev), indexing="ij")
    pix = torch.stack([xs.reshape(-1), ys.reshape(-1), torch.ones(H * W, device=dev)], dim=-1).float()
    Kinv = torch.inverse(K).to(dev)
    light = torch.nn.functional.normalize(torch.tensor([0.4, -0.5, 0.8]), dim=0).to(dev)
    for fi in range(n_frames):
        Rf, Tf = R[fi].to(dev), T[fi].to(dev)
        dcam = pix @ Kinv.T
        dcam = dcam / torch.linalg.norm(dcam, dim=-1, keepdim=True)
        d = dcam @ Rf
        o = (-(Tf @ Rf)).expand_as(d)
        b = (o * d).sum(-1)
        t = (-b - 0.75).clone()                       # start just outside the radius-0.75 ball
        hit = torch.zeros(H * W, dtype=torch.bool, device=dev)
        for _ in range(64):
            p = o + d * t[:, None]
            s = scene_sdf(p)
            hit = s < 5e-4
            t = torch.where(hit, t, t + s.clamp(min=1e-4))
            t = t.clamp(max=6.0)
        p = o + d * t[:, None]
        hit = (scene_sdf(p) < 2e-3) & (t < 5.9)
        n = _normal(p)
        alb = 0.55 + 0.35 * torch.sin(p * torch.tensor([21.0, 17.0, 13.0], device=dev) + torch.tensor([0.0, 1.0, 2.0], device=dev))
        lam = (n * light).sum(-1).clamp(min=0.0)[:, None]
        refl = d - 2 * (d * n).sum(-1, keepdim=True) * n
        spec = (refl * light).sum(-1).clamp(min=0.0)[:, None] ** 16
        col = (alb * (0.25 + 0.75 * lam) + 0.25 * spec).clamp(0.0, 1.0)
        col = torch.where(hit[:, None], col, torch.full_like(col, 0.05))
        ncam = n @ Rf.T
        nu8 = ((ncam * 0.5 + 0.5).clamp(0, 1) * 255.0).round()
        nu8 = torch.where(hit[:, None], nu8, torch.full_like(nu8, 128.0))
        lab = hit.to(torch.int8)
        if hand:
            # seeded ellipse blobs = the occluding hand (labelled -1; object pixels under it are NOT visible)
            hm = torch.zeros(H * W, dtype=torch.bool, device=dev)
            nb = 2
            idx = hit.nonzero().reshape(-1)
            for _ in range(nb):
                if idx.numel() == 0:
                    break
                ci = idx[int(torch.randint(0, idx.numel(), (1,), generator=g).item())]
                cx, cy = float(ci % W), float(ci // W)
                a = W * (0.05 + 0.05 * torch.rand(1, generator=g).item())
                bb = H * (0.03 + 0.04 * torch.rand(1, generator=g).item())
                th = math.pi * torch.rand(1, generator=g).item()
                dx, dy = pix[:, 0] - cx, pix[:, 1] - cy
                u = dx * math.cos(th) + dy * math.sin(th)
                v = -dx * math.sin(th) + dy * math.cos(th)
                hm |= (u / a) ** 2 + (v / bb) ** 2 < 1.0
            lab = torch.where(hm, torch.full_like(lab, -1), lab)
            col = torch.where(hm[:, None], torch.tensor([0.85, 0.62, 0.50], device=dev).expand_as(col), col)
        rgb[fi] = (col * 255.0).round().to(torch.uint8).reshape(H, W, 3)
        label[fi] = lab.reshape(H, W)
        normal[fi] = nu8.to(torch.uint8).reshape(H, W, 3)
    return {"rgb": rgb, "label": label, "normal": normal, "R": R.to(dev), "T": T.to(dev), "K": K.to(dev)}


@torch.no_grad()
def make_correspondences(frames: dict, n_per_pair=2048, offsets=(1, 2, 5), noise_px=0.5, outlier_frac=0.1, seed=99):
    """Synthetic stand-in for the reference's `correspondence_infos` (README.md:43: dense DKM matches): for every frame i and
    partner j = i + offset, pixels of i on the visible object (label 1) are sphere-traced to the analytic surface and
    projected into j; matches whose surface point faces camera j and lands on j's visible object are kept, their partner
    pixel gets Gaussian noise, and a fraction is replaced by gross outliers (random pixel of j) -- what the outlier voting
    is for.  Returns a list of dicts {i, j, kpts0 [M,2] (integer pixel of i, float32), kpts1 [M,2], conf [M]} on the CPU."""
    g = torch.Generator(device="cpu").manual_seed(seed)
    lab = frames["label"]
    dev = lab.device
    F_, H, W = lab.shape
    R, T, K = frames["R"].to(dev), frames["T"].reshape(-1, 3).to(dev), frames["K"].to(dev)
    Kinv = torch.inverse(K)
    out = []
    for i in range(F_):
        idx = (lab[i].reshape(-1) == 1).nonzero().reshape(-1)
        if idx.numel() == 0:
            continue
        for off in offsets:
            j = (i + off) % F_
            if j == i:
                continue
            sel = idx[torch.randint(0, idx.numel(), (n_per_pair,), generator=g).to(dev)]
            px, py = (sel % W).float(), (sel // W).float()
            dcam = torch.stack([px, py, torch.ones_like(px)], -1) @ Kinv.T
            dcam = dcam / torch.linalg.norm(dcam, dim=-1, keepdim=True)
            d = dcam @ R[i]
            o = (-(T[i] @ R[i])).expand_as(d)
            t = (-(o * d).sum(-1) - 0.75).clone()
            for _ in range(64):
                sdf = scene_sdf(o + d * t[:, None])
                t = torch.where(sdf < 5e-4, t, t + sdf.clamp(min=1e-4)).clamp(max=6.0)
            x = o + d * t[:, None]
            n = _normal(x)
            cam_j = -(T[j] @ R[j])
            facing = (n * torch.nn.functional.normalize(cam_j - x, dim=-1)).sum(-1) > 0.15
            y = x @ R[j].T + T[j]
            u = K[0, 0] * y[:, 0] / y[:, 2] + K[0, 2]
            v = K[1, 1] * y[:, 1] / y[:, 2] + K[1, 2]
            ui, vi = u.round().long(), v.round().long()
            inside = (ui >= 0) & (ui < W) & (vi >= 0) & (vi < H) & (y[:, 2] > 0.1)
            vis = torch.zeros_like(inside)
            vis[inside] = lab[j][vi[inside], ui[inside]] == 1
            keep = facing & inside & vis & (scene_sdf(x) < 2e-3)
            k0 = torch.stack([px, py], -1)[keep].cpu()
            k1 = torch.stack([u, v], -1)[keep].cpu()
            m = k0.shape[0]
            if m == 0:
                continue
            k1 = k1 + noise_px * torch.randn(m, 2, generator=g)
            conf = 0.6 + 0.4 * torch.rand(m, generator=g)
            bad = torch.rand(m, generator=g) < outlier_frac
            k1[bad] = torch.stack([torch.rand(int(bad.sum()), generator=g) * (W - 1), torch.rand(int(bad.sum()), generator=g) * (H - 1)], -1)
            conf[bad] = 0.3 + 0.6 * torch.rand(int(bad.sum()), generator=g)
            out.append({"i": i, "j": j, "kpts0": k0.float(), "kpts1": k1.float(), "conf": conf.float(), "is_outlier": bad})
    return out


def write_correspondences_to_disk(matches, dataroot: str, stems=None):
    """<dataroot>/correspondence_infos/<stem_i>_<stem_j>.npz with kpts0 [M,2], kpts1 [M,2] (pixels x, y) and conf [M] -- the
    folder name is the reference's (README.md:43); the file layout inside it is this build's (the reference releases none)."""
    import os
    import numpy as np
    d = os.path.join(dataroot, "correspondence_infos")
    os.makedirs(d, exist_ok=True)
    for m in matches:
        si = stems[m["i"]] if stems else "%04d" % m["i"]
        sj = stems[m["j"]] if stems else "%04d" % m["j"]
        np.savez(os.path.join(d, f"{si}_{sj}.npz"), kpts0=m["kpts0"].numpy(), kpts1=m["kpts1"].numpy(), conf=m["conf"].numpy())


def write_sequence_to_disk(frames: dict, dataroot: str, pose_dir: str | None = None, ext: str = "png"):
    """Write a sequence in the reference's data convention (README.md:27-45; ObjTracker/run.py:74-88,165-179):
    <dataroot>/rgb/%04d.<ext>, sam_seg/%04d.png (channel 1 == 255 object, last channel == 255 hand),
    monocular_normal/%04d.png, and <pose_dir>/%04d.npz with R[3,3], T[1,3], K[3,3]."""
    import os
    import numpy as np
    from PIL import Image
    pose_dir = pose_dir or os.path.join(dataroot, "obj_infos")
    for sub in ("rgb", "sam_seg", "monocular_normal"):
        os.makedirs(os.path.join(dataroot, sub), exist_ok=True)
    os.makedirs(pose_dir, exist_ok=True)
    rgb = frames["rgb"].cpu().numpy(); lab = frames["label"].cpu().numpy(); nrm = frames["normal"].cpu().numpy()
    R = frames["R"].cpu().numpy(); T = frames["T"].cpu().numpy(); K = frames["K"].cpu().numpy()
    for i in range(rgb.shape[0]):
        stem = "%04d" % i
        Image.fromarray(rgb[i]).save(os.path.join(dataroot, "rgb", stem + "." + ext))
        m = np.zeros(rgb[i].shape, np.uint8)
        m[..., 1][lab[i] == 1] = 255
        m[..., 2][lab[i] == -1] = 255
        Image.fromarray(m).save(os.path.join(dataroot, "sam_seg", stem + ".png"))
        Image.fromarray(nrm[i]).save(os.path.join(dataroot, "monocular_normal", stem + ".png"))
        np.savez(os.path.join(pose_dir, stem + ".npz"), R=R[i].astype(np.float32), T=T[i].reshape(1, 3).astype(np.float32),
                 K=K.astype(np.float32))
