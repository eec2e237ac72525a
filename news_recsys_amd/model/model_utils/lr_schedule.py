"""Per-step LR schedule of the reference (src/model/model_utils/lr_schedule.py:6-28):
constant lrs[0] until milestones[0], half-cosine down to lrs[1] at milestones[1], constant after."""
import math

from torch.optim.lr_scheduler import LRScheduler


def cosine_decay_value(step: int, lrs, milestones) -> float:
    lo, hi = milestones[0], milestones[-1]
    if step < lo:
        return lrs[0]
    if step >= hi:
        return lrs[-1]
    frac = (step - lo) / max(1, milestones[1] - lo)
    return lrs[1] + (lrs[0] - lrs[1]) * 0.5 * (1.0 + math.cos(math.pi * frac))


class CosinDecayLR(LRScheduler):
    def __init__(self, optimizer, lrs=(1e-3, 1e-5), milestones=(2000, 5000)):
        self.lrs = list(lrs)
        self.milestones = list(milestones)
        assert len(self.lrs) == 2, "Currently only support 2 lrs for CosinDecayLR"
        assert len(self.lrs) == len(self.milestones), "lrs length must be equal to milestones length"
        super().__init__(optimizer)

    def get_lr(self):
        lr = cosine_decay_value(self.last_epoch, self.lrs, self.milestones)
        return [lr for _ in self.optimizer.param_groups]
