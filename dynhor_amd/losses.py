"""Autograd-path loss stack (torch ops on per-ray tensors) -- same definition as the fused dh_neus_loss kernel:
upstream Runner.train losses with Dynhor's hand gating (keep-mask precedent: reference
ObjTracker/utils/losses.py:69-71) and the MonoSDF-style monocular-normal term."""
import torch


def neus_losses(render_out, true_rgb, obj_mask, keep_mask, igr_weight=0.1, mask_weight=0.1, normal_weight=0.0,
                mono_normal=None, R=None):
    color_fine = render_out["color_fine"]
    m = obj_mask * keep_mask
    mask_sum = m.sum() + 1e-5
    color_loss = ((color_fine - true_rgb) * m).abs().sum() / mask_sum
    psnr = 20.0 * torch.log10(1.0 / (((color_fine - true_rgb) ** 2 * m).sum() / (mask_sum * 3.0)).sqrt())
    eik = render_out["gradient_error"]
    ws = render_out["weight_sum"].clip(1e-3, 1.0 - 1e-3)
    bce = -(obj_mask * torch.log(ws) + (1.0 - obj_mask) * torch.log(1.0 - ws))
    mask_loss = (bce * keep_mask).sum() / (keep_mask.sum() + 1e-5)
    loss = color_loss + igr_weight * eik + mask_weight * mask_loss
    out = {"loss": loss, "color_loss": color_loss, "eikonal_loss": eik, "mask_loss": mask_loss, "psnr": psnr}
    if normal_weight > 0.0 and mono_normal is not None:
        n_obj = (render_out["gradients"] * render_out["weights"][:, :, None]).sum(dim=1)
        n_cam = n_obj @ R.T
        n_hat = n_cam / (torch.linalg.norm(n_cam, dim=-1, keepdim=True) + 1e-6)
        l1 = (n_hat - mono_normal).abs().sum(-1, keepdim=True)
        cs = 1.0 - (n_hat * mono_normal).sum(-1, keepdim=True)
        out["normal_loss"] = ((l1 + cs) * m).sum() / mask_sum
        out["loss"] = loss + normal_weight * out["normal_loss"]
    return out
