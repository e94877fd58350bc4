"""One-cycle learning-rate / beta1 schedule as a pure host-side function of the step index.

Same numbers as ``torch.optim.lr_scheduler.OneCycleLR`` with the reference's settings
(/root/reference/configs/model/maniskill2_act_pcd_model.yaml:16-25: pct_start 0.1, cos anneal,
div_factor 100, final_div_factor 1000; torch defaults cycle_momentum=True, base/max momentum
0.85/0.95 -> AdamW's beta1 is cycled too).  torch's scheduler writes Python floats into the
optimizer's param group, which a captured hipGraph would freeze; this one only *computes* the
values, the trainer writes them into the device-side hyper-parameter array between replays.
"""
import math


class OneCycle:
    def __init__(self, max_lr, total_steps, pct_start=0.3, div_factor=25.0, final_div_factor=1e4,
                 base_momentum=0.85, max_momentum=0.95, cycle_momentum=True):
        if total_steps <= 0:
            raise ValueError("total_steps must be positive")
        self.total_steps = int(total_steps)
        self.max_lr = float(max_lr)
        self.initial_lr = self.max_lr / div_factor
        self.min_lr = self.initial_lr / final_div_factor
        self.base_momentum, self.max_momentum = base_momentum, max_momentum
        self.cycle_momentum = cycle_momentum
        # lr_scheduler.py OneCycleLR.__init__: two phases, end steps as floats
        self.phase1_end = float(pct_start * self.total_steps) - 1
        self.phase2_end = self.total_steps - 1

    @staticmethod
    def _cos(start, end, pct):
        return end + (start - end) / 2.0 * (math.cos(math.pi * pct) + 1)

    def at(self, step_num):
        """(lr, beta1-or-None) used by optimizer step number `step_num` (0-based)."""
        if step_num > self.total_steps:
            raise ValueError(f"Tried to step {step_num} times. The specified number of total steps is {self.total_steps}")
        if step_num <= self.phase1_end:
            pct = step_num / self.phase1_end
            lr = self._cos(self.initial_lr, self.max_lr, pct)
            mom = self._cos(self.max_momentum, self.base_momentum, pct)
        else:
            pct = (step_num - self.phase1_end) / (self.phase2_end - self.phase1_end)
            lr = self._cos(self.max_lr, self.min_lr, pct)
            mom = self._cos(self.base_momentum, self.max_momentum, pct)
        return lr, (mom if self.cycle_momentum else None)
