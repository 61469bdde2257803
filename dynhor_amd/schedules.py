"""Pure host-side schedules of the training loop (upstream Runner) and the data-parallel frame
assignment.  No device code here."""
import math

import torch


def lr_factor(iter_step: int, warm_up_end: int, end_iter: int, alpha: float) -> float:
    """Linear warm-up then cosine decay to alpha (upstream update_learning_rate)."""
    if iter_step < warm_up_end:
        return iter_step / warm_up_end
    progress = (iter_step - warm_up_end) / (end_iter - warm_up_end)
    return (math.cos(math.pi * progress) + 1.0) * 0.5 * (1 - alpha) + alpha


def cos_anneal_ratio(iter_step: int, anneal_end: float) -> float:
    """min(1, iter/anneal_end); 1 if anneal_end == 0 (upstream get_cos_anneal_ratio)."""
    if anneal_end == 0.0:
        return 1.0
    return min(1.0, iter_step / anneal_end)


def frame_slot(iter_step: int, rank: int, world: int) -> int:
    """Index into the shared frame permutation used by `rank` at `iter_step`: ranks take consecutive slots, so one
    iteration covers `world` distinct frames and the ranks never exchange rays (only the gradient)."""
    return iter_step * world + rank


class FramePermutation:
    """The shared frame permutation of the training loop (upstream image_perm, App. A.8): drawn from a seeded CPU
    generator that every rank owns a copy of, re-drawn once per epoch (= n_images slots).  Ranks take consecutive
    slots (frame_slot), so the e-th permutation is the same object on every rank whatever the world size."""

    def __init__(self, n_images: int, seed: int):
        self.n = int(n_images)
        self.gen = torch.Generator(device="cpu")
        self.gen.manual_seed(int(seed))
        self.perm = torch.randperm(self.n, generator=self.gen)
        self.epoch = 0

    def frame(self, slot: int) -> int:
        epoch = slot // self.n
        if epoch < self.epoch:
            raise ValueError("FramePermutation only moves forward (slot belongs to a past epoch)")
        while self.epoch < epoch:
            self.perm = torch.randperm(self.n, generator=self.gen)
            self.epoch += 1
        return int(self.perm[slot % self.n])

    def state_dict(self):
        return {"gen": self.gen.get_state(), "perm": self.perm.clone(), "epoch": self.epoch}

    def load_state_dict(self, sd):
        self.gen.set_state(sd["gen"].cpu())
        self.perm = sd["perm"].cpu().clone()
        self.epoch = int(sd["epoch"])
