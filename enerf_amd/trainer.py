"""Minimal train / render harness around the hot path (the bench's "step"), re-stating what main_nerf.py +
Trainer.train_one_epoch do per iteration in the reference (main_nerf.py:211-214, nerf/utils.py:943-997):

    every 16 steps: model.update_extra_state()          (density-grid EMA + packbits + mean_count)
    render -> loss -> backward -> [all-reduce grads] -> Adam(betas=(0.9, 0.99), eps=1e-15) step

`occupancy="synthetic"` keeps marching against the analytic scene's bitfield: update_extra_state() still runs (its cost
is part of the step) but the synthetic grid is restored afterwards, so sample counts stay reproducible with
random-init weights.
"""
import torch

from . import scene
from .optim import FusedAdam
from .parallel import GradAverager


class TrainHarness:
    def __init__(self, model, lr=1e-2, occupancy="synthetic", world=1, update_interval=16):
        self.model = model
        adam = FusedAdam if next(model.parameters()).is_cuda else torch.optim.Adam
        self.opt = adam(model.get_params(lr), betas=(0.9, 0.99), eps=1e-15)
        self.occupancy = occupancy
        self.update_interval = update_interval
        self.global_step = 0
        self.avg = GradAverager(list(model.parameters())) if world > 1 else None
        self._syn = None
        if model.cuda_ray and occupancy == "synthetic":
            self._syn = scene.install_occupancy(model)

    def maybe_update_extra_state(self):
        m = self.model
        if m.cuda_ray and self.global_step % self.update_interval == 0:
            m.update_extra_state()
            if self._syn is not None:
                m.density_grid.copy_(self._syn[0])
                m.density_bitfield.copy_(self._syn[1])

    def step_rgb(self, rays_o, rays_d, target, **render_kw):
        """One RGB training step (nerf/utils.py:575-640 train_step + the optimizer part of train_one_epoch)."""
        self.model.train()
        self.maybe_update_extra_state()
        self.global_step += 1
        self.opt.zero_grad(set_to_none=True)
        out = self.model.render(rays_o, rays_d, staged=False, bg_color=None, perturb=True, **render_kw)
        loss = torch.nn.functional.mse_loss(out["image"], target)
        loss.backward()
        if self.avg is not None:
            self.avg()
        self.opt.step()
        return loss

    def step_events(self, data, opt):
        """One event training step: two renders sharing one backward (nerf/utils.py:482-573)."""
        from .events import train_step_events
        self.model.train()
        self.maybe_update_extra_state()
        self.global_step += 1
        self.opt.zero_grad(set_to_none=True)
        loss, _ = train_step_events(self.model, data, opt)
        loss.backward()
        if self.avg is not None:
            self.avg()
        self.opt.step()
        return loss
