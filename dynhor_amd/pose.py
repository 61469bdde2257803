"""Per-frame object-pose refinement: the stage-1 poses (ObjTracker/run.py:165-179: R = rot6d_to_matrix(.)^T
object -> camera, T) become trainable cameras of the reconstruction stage.  Parameters follow the reference's own pose
model -- a 6-D rotation per frame (ObjTracker/utils/geometry.py:7-25, Zhou et al.) and a translation -- and its optimiser
convention: Adam with the rotation at 10x the learning rate of the rest (ObjTracker/jointopt.py:125-141).

The heavy part -- d loss / d rays through both MLPs incl. the second-order path -- is HIP (dh_color_backward_rays,
dh_sdf_backward_rays, dh_render_scan_bwd_rays); this module only chains those per-ray adjoints [B,3] into the 9 pose numbers
of the frame with torch autograd over the ray-generation formula (a few [B,3] ops)."""
from __future__ import annotations

import torch


def rot6d_to_matrix(x: torch.Tensor) -> torch.Tensor:
    """[F,3,2] -> [F,3,3]: columns a1, a2 -> Gram-Schmidt -> stack(b1, b2, b3, dim=-1)   (utils/geometry.py:7-25)."""
    a1, a2 = x[:, :, 0], x[:, :, 1]
    b1 = torch.nn.functional.normalize(a1, dim=-1)
    b2 = torch.nn.functional.normalize(a2 - (b1 * a2).sum(-1, keepdim=True) * b1, dim=-1)
    b3 = torch.linalg.cross(b1, b2, dim=-1)
    return torch.stack((b1, b2, b3), dim=-1)


class PoseRefiner(torch.nn.Module):
    def __init__(self, R0: torch.Tensor, T0: torch.Tensor, lr=1e-4, rot_lr_mult=10.0):
        super().__init__()
        # saved R = rot6d_to_matrix(r6)^T  =>  r6 = first two columns of R^T (utils/geometry.py:28-38 matrix_to_rot6d)
        self.rot6d = torch.nn.Parameter(R0.transpose(1, 2)[:, :, :2].clone().float())
        self.trans = torch.nn.Parameter(T0.reshape(-1, 3).clone().float())
        self.opt = torch.optim.Adam([{"params": [self.rot6d], "lr": lr * rot_lr_mult}, {"params": [self.trans], "lr": lr}])

    def poses(self):
        """(R [F,3,3] object -> camera, T [F,3]) in the layout of obj_infos/*.npz."""
        return rot6d_to_matrix(self.rot6d).transpose(1, 2), self.trans

    def rays(self, frame: int, px: torch.Tensor, py: torch.Tensor, Kinv: torch.Tensor):
        """Differentiable rays of `frame` through pixels (px, py): o = -R^T T, d = R^T normalize(K^-1 [u, v, 1])   (App. B)."""
        R = rot6d_to_matrix(self.rot6d[frame:frame + 1])[0].T
        T = self.trans[frame]
        pix = torch.stack([px.float(), py.float(), torch.ones_like(px, dtype=torch.float32)], -1)
        dc = torch.nn.functional.normalize(pix @ Kinv.T, dim=-1)
        d = dc @ R                      # R^T dc per row
        o = (-(T @ R)).expand_as(d)
        return o, d, R

    def step(self, o, d, R, d_rays_o, d_rays_d, d_R=None, grad_scale=1.0, allreduce=None, partner=None):
        """Chain the HIP path's per-ray adjoints into the pose parameters and take one Adam step.  `allreduce` (data-parallel
        runs): called on each parameter gradient before the step -- every rank trains a different frame, the sum gives all
        ranks the same pose update (two tiny [F,3,2] / [F,3] buffers).  partner = (d_R_all [F,3,3], d_T_all [F,3]): the
        correspondence term's gradient w.r.t. EVERY frame's saved pose (it projects into partner frames)."""
        self.opt.zero_grad(set_to_none=True)
        outs, grads = [o, d], [d_rays_o, d_rays_d]
        if d_R is not None:
            outs.append(R); grads.append(d_R)
        if partner is not None:
            R_all, T_all = self.poses()
            outs += [R_all, T_all]; grads += [partner[0], partner[1].reshape(T_all.shape)]
        torch.autograd.backward(outs, grads)
        for p in (self.rot6d, self.trans):
            if allreduce is not None:
                allreduce(p.grad)
            if grad_scale != 1.0:
                p.grad.mul_(grad_scale)
        self.opt.step()
