"""Runner -- mirror of upstream ``exp_runner.py:Runner`` in the reference's config style
(flat yaml.safe_load dict, ObjTracker/configs/custom_shoes.yaml:1-18; experiment directory exps/<seq>/<exp>/ with the
config copied in, ObjTracker/run.py:125-128; one scalar per loss key per step as jointopt.py:151-153 logs them).

Data-parallel: one process per GPU (torch.distributed, backend nccl == RCCL over xGMI).  Rank r takes frame
perm[(iter*world + r) % n_images]; the only exchange is ONE all-reduce of the flat 802,491-float gradient per
iteration, then every rank applies the same fused Adam step.
"""
from __future__ import annotations

import json
import math
import os
import shutil

import numpy as np
import torch
import torch.distributed as dist
import yaml

from . import dist as dh_dist
from . import schedules
from .dataset import Dataset
from .fields import ParamStore, RenderingNetwork, SDFNetwork, SingleVarianceNetwork
from .renderer import NeuSRenderer

DEFAULT_CONF = {
    "seq_name": "synthetic", "exp_name": "neus",
    "data_info": {"dataroot": None, "synthetic": {"n_frames": 64, "H": 512, "W": 512, "seed": 4321}},
    "train": {"learning_rate": 5e-4, "learning_rate_alpha": 0.05, "end_iter": 300000, "batch_size": 2048,
              "warm_up_end": 5000, "anneal_end": 50000, "igr_weight": 0.1, "mask_weight": 0.1, "normal_weight": 0.0,
              "save_freq": 10000, "val_freq": 2500, "report_freq": 100, "use_white_bkgd": False, "keep_only": False,
              "seed": 1234, "ray_seed": 4321,
              # roctx: put every C-ABI stage call into a ROCm marker range (rocprofv3 --marker-trace); off by default
              "roctx": False,
              # full loss stack (BASELINE.json configs[4]): dense-correspondence reprojection term, DESIGN.md section 9
              "corr_weight": 0.0, "corr_fraction": 0.25, "corr_delta_px": 4.0, "corr_vote_freq": 0, "corr_vote_tau_px": 8.0,
              # per-frame pose refinement: 6-D rotation + translation per frame, rotation at 10x lr
              "refine_poses": False, "pose_lr": 1e-4, "pose_rot_lr_mult": 10.0, "pose_start_iter": 0},
    # family "neus": the 8x256 / 4x256 fp32 MLPs (BASELINE.json configs[1]); "hash": hash-grid + shallow MLPs (configs[3])
    # arithmetic: null (the library's default: split_f16), "split_f16" (two fp16 pieces / three products, shipping), "split_bf16"
    # (three bf16 pieces / six products) or "fp32_mfma" (the native fp32-MFMA twins) for THIS Runner's renderer -- passed with every
    # launch, so Runners of one process (also on different host threads) may differ
    "model": {"family": "neus", "arithmetic": None, "sdf_network": {}, "variance_network": {"init_val": 0.3}, "rendering_network": {},
              "hash_sdf_network": {}, "sh_rendering_network": {},
              # hash family only: sampler "hierarchical" (NeuS 64+64) or "occgrid" (instant-nsr-pl occupancy-grid marching);
              # reproducible_table_grad: the table scatter through fixed-point integer atomics (bitwise reproducible training; False:
              # float atomics)
              "hash_renderer": {"sampler": "hierarchical", "march_samples_per_ray": 512, "grid_res": 128, "grid_update_every": 16,
                                "max_samples": 128, "reproducible_table_grad": True},
              "neus_renderer": {"n_samples": 64, "n_importance": 64, "n_outside": 0, "up_sample_steps": 4, "perturb": 1.0}},
}


def _merge(base, over):
    out = dict(base)
    for k, v in (over or {}).items():
        out[k] = _merge(base[k], v) if isinstance(v, dict) and isinstance(base.get(k), dict) else v
    return out


class Runner:
    def __init__(self, conf_path=None, mode="train", case=None, is_continue=False, conf: dict | None = None,
                 dataset: Dataset | None = None, device=None, exp_root="exps"):
        if conf is None:
            with open(conf_path, "r") as f:
                conf = yaml.safe_load(f)
        self.conf = _merge(DEFAULT_CONF, conf)
        if case is not None:
            self.conf["seq_name"] = case
        self.mode = mode
        self.world = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
        self.rank = dist.get_rank() if self.world > 1 else 0
        self.device = torch.device(device if device is not None else f"cuda:{int(os.environ.get('LOCAL_RANK', 0))}")
        self.base_exp_dir = os.path.join(exp_root, self.conf["seq_name"], self.conf["exp_name"])
        if self.rank == 0:
            os.makedirs(self.base_exp_dir, exist_ok=True)
            if conf_path is not None:
                shutil.copy(conf_path, os.path.join(self.base_exp_dir, "config.yaml"))
            else:
                with open(os.path.join(self.base_exp_dir, "config.yaml"), "w") as f:
                    yaml.safe_dump(self.conf, f)
        tr = self.conf["train"]
        self.end_iter = tr["end_iter"]; self.batch_size = tr["batch_size"]
        self.learning_rate = tr["learning_rate"]; self.learning_rate_alpha = tr["learning_rate_alpha"]
        self.warm_up_end = tr["warm_up_end"]; self.anneal_end = tr["anneal_end"]
        self.igr_weight = tr["igr_weight"]; self.mask_weight = tr["mask_weight"]; self.normal_weight = tr["normal_weight"]
        self.save_freq = tr["save_freq"]; self.val_freq = tr["val_freq"]; self.report_freq = tr["report_freq"]
        self.use_white_bkgd = tr["use_white_bkgd"]; self.keep_only = tr["keep_only"]
        self.corr_weight = tr["corr_weight"]; self.corr_fraction = tr["corr_fraction"]; self.corr_delta_px = tr["corr_delta_px"]
        self.corr_vote_freq = tr["corr_vote_freq"]; self.corr_vote_tau_px = tr["corr_vote_tau_px"]
        self.iter_step = 0

        if dataset is None:
            di = self.conf["data_info"]
            if di.get("dataroot"):
                dataset = Dataset(di, device=self.device)
            else:
                dataset = Dataset.from_synthetic(device=self.device, **di["synthetic"])
        self.dataset = dataset

        with torch.random.fork_rng(devices=[]):
            torch.manual_seed(tr["seed"])          # identical initial weights on every rank
            family = self.conf["model"]["family"]
            if family == "neus":
                self.sdf_network = SDFNetwork(**self.conf["model"]["sdf_network"])
                self.color_network = RenderingNetwork(**self.conf["model"]["rendering_network"])
                store_cls, renderer_cls = ParamStore, NeuSRenderer
            elif family == "hash":
                from .hash_fields import HashNeuSRenderer, HashParamStore, HashSDFNetwork, SHRenderingNetwork
                self.sdf_network = HashSDFNetwork(**self.conf["model"]["hash_sdf_network"])
                self.color_network = SHRenderingNetwork(**self.conf["model"]["sh_rendering_network"])
                store_cls, renderer_cls = HashParamStore, HashNeuSRenderer
            else:
                raise ValueError(f"model.family must be 'neus' or 'hash', got {family!r}")
            self.deviation_network = SingleVarianceNetwork(**self.conf["model"]["variance_network"])
        self.nerf_outside = None
        self.store = store_cls(self.sdf_network, self.deviation_network, self.color_network, self.device)
        extra = dict(self.conf["model"]["hash_renderer"]) if family == "hash" else {}
        ar = self.conf["model"].get("arithmetic")
        if ar is not None:
            from . import _lib
            extra["arithmetic"] = _lib.ARITH_NAMES[ar]
        self.renderer = renderer_cls(self.nerf_outside, self.sdf_network, self.deviation_network, self.color_network,
                                     store=self.store, device=self.device, **self.conf["model"]["neus_renderer"], **extra)
        if tr.get("roctx"):
            self.renderer.timer.set_markers(True)
        # hash family, data-parallel: the 49 MB table gradient is reduced by a collective of its own, started the moment the table
        # scatter has been enqueued and overlapped with the small weight-gradient GEMMs (hash_fields.HashNeuSRenderer._weight_grads);
        # train.overlap_table_reduce = False keeps the one serial all-reduce of the whole bucket (the two are bitwise identical at
        # two ranks; at more ranks a ring may add the same numbers in a different order)
        self.overlap_table_reduce = bool(tr.get("overlap_table_reduce", True)) and hasattr(self.store, "table_floats") \
            and dist.is_available() and dist.is_initialized()
        if self.overlap_table_reduce:
            self.renderer.table_grad_hook = lambda t: dist.all_reduce(t, op=dist.ReduceOp.SUM, async_op=True)
        self.pose_refiner = None
        if tr["refine_poses"]:
            from .pose import PoseRefiner
            self.pose_refiner = PoseRefiner(self.dataset.R, self.dataset.T, lr=tr["pose_lr"], rot_lr_mult=tr["pose_rot_lr_mult"]).to(self.device)
            self.pose_start_iter = tr["pose_start_iter"]
        self.ray_gen = torch.Generator(device=self.device)
        self.ray_gen.manual_seed(tr["ray_seed"] * 1000003 + self.rank)
        self.frame_perm = schedules.FramePermutation(self.dataset.n_images, tr["ray_seed"])   # same on every rank
        self.scalars = []
        self._board = None
        if is_continue:
            ck_dir = os.path.join(self.base_exp_dir, "checkpoints")
            ck = sorted(f for f in os.listdir(ck_dir) if f.endswith(".pth")) if os.path.isdir(ck_dir) else []
            if ck:
                self.load_checkpoint(ck[-1])

    # ------------------------------------------------------------------ schedules (App. A.8)
    @property
    def image_perm(self):
        """Upstream name of the current epoch's frame permutation."""
        return self.frame_perm.perm

    def get_cos_anneal_ratio(self):
        return schedules.cos_anneal_ratio(self.iter_step, self.anneal_end)

    def current_lr(self):
        return self.learning_rate * schedules.lr_factor(self.iter_step, self.warm_up_end, self.end_iter,
                                                        self.learning_rate_alpha)

    def update_learning_rate(self):
        self.lr = self.current_lr()
        return self.lr

    def frame_for_slot(self, slot: int) -> int:
        """Frame of permutation slot `slot`: the permutation is re-drawn once per epoch (upstream re-permutes when
        iter_step % n_images == 0) from the SHARED perm_gen, so every rank holds the same e-th permutation for epoch e
        whatever the world size."""
        return self.frame_perm.frame(slot)

    # ------------------------------------------------------------------ one iteration (the hot loop)
    def train_iteration(self):
        frame = self.frame_for_slot(schedules.frame_slot(self.iter_step, self.rank, self.world))
        refine = self.pose_refiner is not None and self.iter_step >= self.pose_start_iter
        corr = None
        if self.corr_weight > 0.0 and self.dataset.corr is not None:
            if self.corr_vote_freq and self.iter_step > 0 and self.iter_step % self.corr_vote_freq == 0:
                self.vote_correspondences()
            rays, corr, _ = self.dataset.gen_corr_rays_at(frame, self.batch_size, int(self.batch_size * self.corr_fraction),
                                                         generator=self.ray_gen)
        else:
            rays = self.dataset.gen_random_rays_at(frame, self.batch_size, keep_only=self.keep_only, generator=self.ray_gen)
        near, far = self.dataset._last_near_far
        self._last_rays = rays
        bg = torch.ones(3, device=self.device) if self.use_white_bkgd else None
        # the per-ray perturbation of the coarse samples comes from the SAME checkpointed stream as the pixels
        t_rand = torch.rand([rays.shape[0], 1], device=self.device, generator=self.ray_gen) if self.renderer.perturb > 0 else None
        stats = self.renderer.train_step_core(rays, near, far, self.dataset.R[frame], self.get_cos_anneal_ratio(),
                                              self.igr_weight, self.mask_weight, self.normal_weight, background_rgb=bg,
                                              t_rand=t_rand, corr=corr, corr_weight=self.corr_weight,
                                              corr_frames=self.dataset.corr_frames() if corr is not None else None,
                                              corr_delta_px=self.corr_delta_px, ray_grads=refine)
        if refine:
            # chain the per-ray adjoints into this frame's 9 pose numbers (torch autograd over the ray formula), step, and
            # write the refined pose back into the resident frame table the HIP ray gather reads
            px, py = self.dataset._last_pixels
            o, d, Rf = self.pose_refiner.rays(frame, px, py, self.dataset.Kinv)
            d_o, d_d, d_R = self.renderer.last_ray_grads
            self.pose_refiner.step(o, d, Rf, d_o, d_d, d_R, grad_scale=1.0 / self.world,
                                   allreduce=dh_dist.allreduce_sum_ if self.world > 1 else None,
                                   partner=getattr(self.renderer, "last_partner_pose_grads", None))
            with torch.no_grad():
                Rn, Tn = self.pose_refiner.poses()
                self.dataset.R.copy_(Rn); self.dataset.T.copy_(Tn)
        grad = self.store.grad_flat
        pending = getattr(self.renderer, "pending_table_reduce", None)
        if pending is not None:
            # hash family: the table slice is already on its way (started behind the table scatter); reduce the small rest, then join
            dh_dist.allreduce_sum_(grad[self.store.table_floats:])
            pending.wait()
            self.renderer.pending_table_reduce = None
        else:
            dh_dist.allreduce_sum_(grad)                         # RCCL over xGMI: one 3.2 MB bucket
        lr = self.current_lr()          # 0 at iter_step 0, as upstream (update_learning_rate() runs before the loop)
        self.store.adam_step(lr, grad=grad, grad_scale=1.0 / self.world)
        self.iter_step += 1
        return stats

    def train(self, n_iters=None):
        end = self.end_iter if n_iters is None else min(self.end_iter, self.iter_step + n_iters)
        while self.iter_step < end:
            stats = self.train_iteration()
            if self.iter_step % self.report_freq == 0:
                self.report(stats)
            if self.rank == 0 and self.iter_step % self.save_freq == 0:
                self.save_checkpoint()
            if self.rank == 0 and self.val_freq and self.iter_step % self.val_freq == 0:
                self.validate_image()
        self.close()
        return self

    def close(self):
        """Flush and close the scalar writer (train() calls it; safe to call twice)."""
        if self._board is not None:
            self._board.flush()
            self._board.close()
            self._board = None

    def report(self, stats):
        v = dh_dist.mean_stats(stats).tolist()
        # loud, never silent: the two-piece fp16 arithmetic's range watch (one device read, report iterations
        # only) and a finite-loss check
        # (reduced with MAX over the ranks first -- ADVICE r5: a rank that raised alone left the others in the next all-reduce)
        if hasattr(self.renderer, "check_range"):
            self.renderer.check_range(max_over_ranks=lambda x: dh_dist.max_over_ranks(x, self.device))
        if not math.isfinite(v[0]):
            from ._lib import DynhorHipError
            raise DynhorHipError(f"non-finite loss at iteration {self.iter_step}: {v[0]}")
        rec = {"iter": self.iter_step, "Loss/loss": v[0], "Loss/color_loss": v[1], "Loss/eikonal_loss": v[2],
               "Loss/mask_loss": v[3], "Loss/normal_loss": v[4], "Statistics/psnr": v[5],
               **({"Loss/corr_loss": float(self.renderer.last_corr_stats[0]), "Statistics/corr_residual_px": float(self.renderer.last_corr_stats[2])}
                  if self.corr_weight > 0.0 and getattr(self.renderer, "last_corr_stats", None) is not None else {}),
               "Statistics/s_val": float(1.0 / self.store.inv_s().item()), "lr": self.current_lr()}
        st = getattr(self.renderer, "last_state", None)
        if st is not None and getattr(self, "_last_rays", None) is not None:
            # upstream Statistics/cdf and Statistics/weight_max (App. A.8), masked by obj*keep like the colour loss
            m = (self._last_rays[:, 9:10] * self._last_rays[:, 10:11])
            msum = float(m.sum()) + 1e-5
            if st.cdf.dim() == 2:
                cdf0 = st.cdf[:, :1]
            else:
                # packed rays (hash family, occupancy-grid sampler): cdf is the flat capacity buffer; a ray's first sample sits at
                # its segment offset, rays without samples contribute nothing (st.cdf[:, :1] raised IndexError here)
                off, cnt = st.m.off, st.m.cnt
                cdf0 = (st.cdf[off.clamp(max=st.cdf.shape[0] - 1)] * (cnt > 0)).view(-1, 1)
            rec["Statistics/cdf"] = float((cdf0 * m).sum()) / msum
            rec["Statistics/weight_max"] = float((st.wmax * m).sum()) / msum
        if self.rank == 0:
            self.scalars.append(rec)
            with open(os.path.join(self.base_exp_dir, "scalars.jsonl"), "a") as f:
                f.write(json.dumps(rec) + "\n")
            # the reference's logger: one scalar per key per step to a SummaryWriter under <exp>/board
            # (ObjTracker/run.py:125-127, jointopt.py:151-153)
            if self._board is None:
                from .tb_events import make_writer
                self._board = make_writer(os.path.join(self.base_exp_dir, "board"))
            for k, v in rec.items():
                if k != "iter":
                    self._board.add_scalar(k, v, self.iter_step)
            self._board.flush()
        return rec

    @torch.no_grad()
    def vote_correspondences(self, chunk=8192):
        """Outlier voting over ALL matches with the current geometry (forward-only render of every matched ray, residuals from
        dh_corr_loss, then Dataset.vote_correspondences).  Every rank evaluates the same matches with the same weights, so
        the certainties stay identical across ranks without a collective."""
        from . import _lib
        from .renderer import _p
        ds, ren = self.dataset, self.renderer
        res = torch.full((ds.corr.shape[0],), float("nan"), device=self.device)
        R_all, T_all, K = ds.corr_frames()
        for f, (lo, hi) in ds._corr_range.items():
            for s0 in range(lo, hi, chunk):
                m = ds.corr[s0:min(s0 + chunk, hi)]
                B = m.shape[0]
                rays = ds.gen_rays_at_pixels(f, m[:, 0].long(), m[:, 1].long())
                near, far = ds._last_near_far
                o, d = rays[:, :3].contiguous(), rays[:, 3:6].contiguous()
                z = ren.sample_z(o, d, near, far, perturb_overwrite=0)
                st = ren._forward_core(o, d, z, self.get_cos_anneal_ratio(), None, want_nmap=False, infer_only=True)
                corr = torch.cat([m[:, 2:4], torch.ones(B, 1, device=self.device), m[:, 5:6]], -1).contiguous()
                cst = torch.empty(4, device=self.device); r = torch.empty(B, device=self.device)
                dw = torch.empty(B, st.n, device=self.device)
                _lib.check(_lib.lib().dh_corr_loss(_p(o), _p(d), _p(z), _p(st.weights), _p(corr), _p(R_all), _p(T_all),
                                                  int(R_all.shape[0]), _p(K), B, st.n, st.sample_dist, float(self.corr_delta_px), 1.0,
                                                  _p(cst), _p(r), _p(dw), None, _lib.stream()))
                # rays that render (almost) nothing have no surface estimate yet: leave them un-voted (NaN); a match whose point
                # lands behind the partner camera comes back +inf and votes as an outlier
                res[s0:s0 + B] = torch.where(st.wsum.view(-1) > 0.5, r, torch.full_like(r, float("nan")))
        self.last_vote = ds.vote_correspondences(res, tau_px=self.corr_vote_tau_px)
        return self.last_vote

    # ------------------------------------------------------------------ checkpoints (App. A.8 layout)
    def save_checkpoint(self):
        ck = {"nerf": {}, "sdf_network_fine": self.sdf_network.state_dict(),
              "variance_network_fine": self.deviation_network.state_dict(),
              "color_network_fine": self.color_network.state_dict(),
              "optimizer": self.store.optimizer_state_dict(self.current_lr()), "iter_step": self.iter_step,
              # extra keys (upstream loaders ignore them): the ray / permutation RNG streams, so that a resumed run draws
              # the pixels and frames an uninterrupted one would have drawn
              "dynhor_rng": {"ray_gen": self.ray_gen.get_state(), "frame_perm": self.frame_perm.state_dict(),
                             "rank": self.rank, "world": self.world}}
        if self.pose_refiner is not None:
            ck["pose_refiner"] = {"model": self.pose_refiner.state_dict(), "optimizer": self.pose_refiner.opt.state_dict()}
        occ = getattr(self.renderer, "sampler_state_dict", lambda: None)()
        if occ is not None:
            ck["dynhor_occgrid"] = occ
        d = os.path.join(self.base_exp_dir, "checkpoints")
        os.makedirs(d, exist_ok=True)
        path = os.path.join(d, "ckpt_{:0>6d}.pth".format(self.iter_step))
        torch.save({k: (v if not isinstance(v, dict) else _to_cpu(v)) for k, v in ck.items()}, path)
        return path

    def load_checkpoint(self, checkpoint_name):
        path = checkpoint_name if os.path.isabs(checkpoint_name) or os.path.exists(checkpoint_name) else \
            os.path.join(self.base_exp_dir, "checkpoints", checkpoint_name)
        ck = torch.load(path, map_location=self.device, weights_only=False)
        self.sdf_network.load_state_dict(ck["sdf_network_fine"])
        self.deviation_network.load_state_dict(ck["variance_network_fine"])
        self.color_network.load_state_dict(ck["color_network_fine"])
        opt = ck.get("optimizer")
        if opt:
            st = opt.get("state", {})
            extra = len(st) - len(self.store.slices)
            if extra > 0:   # upstream checkpoints list the (unused here) NeRF++ background parameters first
                opt = {"state": {i - extra: v for i, v in st.items() if i >= extra}, "param_groups": opt["param_groups"]}
            self.store.load_optimizer_state_dict(opt)
        self.iter_step = ck["iter_step"]
        if self.pose_refiner is not None and "pose_refiner" in ck:
            self.pose_refiner.load_state_dict(ck["pose_refiner"]["model"])
            self.pose_refiner.opt.load_state_dict(ck["pose_refiner"]["optimizer"])
            with torch.no_grad():
                Rn, Tn = self.pose_refiner.poses()
                self.dataset.R.copy_(Rn); self.dataset.T.copy_(Tn)
        if "dynhor_occgrid" in ck and hasattr(self.renderer, "load_sampler_state_dict"):
            self.renderer.load_sampler_state_dict(ck["dynhor_occgrid"])
        rng = ck.get("dynhor_rng")
        if rng is not None:
            self.frame_perm.load_state_dict(rng["frame_perm"])
            if rng.get("rank", 0) == self.rank and rng.get("world", 1) == self.world:
                self.ray_gen.set_state(rng["ray_gen"].cpu())
            else:   # only rank 0 writes checkpoints: the other ranks restart their own stream at a point no earlier
                    # iteration used (not bit-equal to an uninterrupted run, but no pixel draw is replayed)
                self.ray_gen.manual_seed((self.conf["train"]["ray_seed"] * 1000003 + self.rank) ^ (self.iter_step * 2654435761 % (1 << 31)))
        self.store.bump()

    # ------------------------------------------------------------------ validation
    @torch.no_grad()
    def render_image(self, idx, resolution_level=4, chunk=4096):
        rays, h, w = self.dataset.gen_rays_at(idx, resolution_level)
        near, far = self.dataset._last_near_far
        cols, nrms = [], []
        bg = torch.ones(3, device=self.device) if self.use_white_bkgd else None
        for s in range(0, rays.shape[0], chunk):
            r = rays[s:s + chunk]
            o, d = r[:, :3].contiguous(), r[:, 3:6].contiguous()
            c, nm = self.renderer.render_rays(o, d, near[s:s + chunk], far[s:s + chunk], self.get_cos_anneal_ratio(), bg)
            cols.append(c); nrms.append(nm)
        return torch.cat(cols).view(h, w, 3), torch.cat(nrms).view(h, w, 3), rays.view(h, w, 14)

    @torch.no_grad()
    def validate_image(self, idx=-1, resolution_level=4):
        if idx < 0:
            idx = int(np.random.randint(self.dataset.n_images))
        img, nrm, rays = self.render_image(idx, resolution_level)
        m = (rays[..., 9:10] * rays[..., 10:11])
        mse = (((img - rays[..., 6:9]) ** 2) * m).sum() / (m.sum() * 3.0 + 1e-5)
        psnr = float(20.0 * torch.log10(1.0 / mse.sqrt()))
        if self.rank == 0:
            from PIL import Image
            d = os.path.join(self.base_exp_dir, "validations_fine")
            dn = os.path.join(self.base_exp_dir, "normals")
            os.makedirs(d, exist_ok=True); os.makedirs(dn, exist_ok=True)
            gt = (rays[..., 6:9].clamp(0, 1) * 255).byte().cpu().numpy()
            pr = (img.clamp(0, 1) * 255).byte().cpu().numpy()
            # upstream writes prediction and ground truth side by side
            Image.fromarray(np.concatenate([pr, gt], axis=1)).save(os.path.join(d, "{:0>8d}_{}.png".format(self.iter_step, idx)))
            n_cam = nrm @ self.dataset.R[idx].T                      # object -> camera frame (x_cam = R x_obj)
            nimg = ((n_cam / (n_cam.norm(dim=-1, keepdim=True) + 1e-6)) * 0.5 + 0.5).clamp(0, 1)
            Image.fromarray((nimg * 255).byte().cpu().numpy()).save(os.path.join(dn, "{:0>8d}_{}.png".format(self.iter_step, idx)))
        return psnr

    @torch.no_grad()
    def validate_mesh(self, resolution=64, threshold=0.0, world_space=False, save=True):
        """Upstream Runner.validate_mesh / NeuSRenderer.extract_geometry: -sdf on a regular grid over
        the object bounding box (HIP no-grad SDF kernel, 64^3-point chunks), iso-surface by marching cubes (model.mesh_method: 'cubes' | 'tetrahedra')
        (dynhor_amd/mesh.py; mcubes is not available), written as meshes/<iter>.ply.  Returns (vertices, triangles)."""
        from .mesh import write_ply
        bmin, bmax = self.dataset.object_bbox_min, self.dataset.object_bbox_max
        verts, faces = self.renderer.extract_geometry(bmin, bmax, resolution=resolution, threshold=threshold,
                                                      method=self.conf.get("model", {}).get("mesh_method", "cubes"))
        if save and self.rank == 0:
            d = os.path.join(self.base_exp_dir, "meshes")
            os.makedirs(d, exist_ok=True)
            write_ply(os.path.join(d, "{:0>8d}.ply".format(self.iter_step)), verts, faces)
        return verts, faces


def _to_cpu(d):
    return {k: (_to_cpu(v) if isinstance(v, dict) else (v.detach().cpu() if torch.is_tensor(v) else v)) for k, v in d.items()}
