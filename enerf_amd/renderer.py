"""`NeRFRenderer`: the module the networks derive from, i.e. the caller of the hot path.

What is kept from the reference (nerf/renderer.py:86-126, 566-599) is the contract other code depends on:
constructor arguments; the registered buffers and their shapes -- `aabb_train`, `aabb_infer` [6], `density_grid`
[cascade, 128^3] fp32, `density_bitfield` [cascade * 128^3 / 8] u8, `step_counter` [16, 2] i32 -- so that a reference
checkpoint's state_dict loads; the host-side counters (`mean_density`, `iter_density`, `mean_count`, `local_step`); and
the keyword interface of `render` / `run` / `run_cuda` / `update_extra_state` / `mark_untrained_grid`.

The bodies live where the work is done:

    run                  sampler.render_stratified        pure-PyTorch sampler (cuda_ray off; cpu_baseline)
    run_cuda, training   fused_render.render_train        one autograd node per render (MI355X fp32 path)
                         _train_ops                       the four autograd Functions one after the other
    run_cuda, eval       frame.render_frame               every sample marched at once, one compositing pass
                         frame.render_rounds              the reference's round schedule
    update_extra_state   density_update.update / update_torch
    mark_untrained_grid  density_update.mark_untrained / mark_untrained_torch
"""
import math

import torch
import torch.nn as nn

from . import density_update, frame, fused_render, raymarching, sampler


class NeRFRenderer(nn.Module):
    grid_size = 128

    def __init__(self, bound=1, cuda_ray=False, density_scale=1, min_near=0.2, density_thresh=0.01, bg_radius=-1):
        super().__init__()
        self.bound, self.cuda_ray, self.bg_radius = bound, cuda_ray, bg_radius
        self.density_scale, self.density_thresh, self.min_near = density_scale, density_thresh, min_near
        self.cascade = 1 + int(math.ceil(math.log2(bound)))       # occupancy levels: cubes of half-width 1, 2, 4, ... >= bound
        cube = torch.tensor([-bound] * 3 + [bound] * 3, dtype=torch.float32)
        self.register_buffer("aabb_train", cube)
        self.register_buffer("aabb_infer", cube.clone())
        if cuda_ray:
            cells = self.grid_size ** 3
            self.register_buffer("density_grid", torch.zeros(self.cascade, cells))
            self.register_buffer("density_bitfield", torch.zeros(self.cascade * cells // 8, dtype=torch.uint8))
            self.register_buffer("step_counter", torch.zeros(16, 2, dtype=torch.int32))      # (samples, rays) per render
            self._zero_counters()

    def _zero_counters(self):
        self.mean_density = self.iter_density = self.mean_count = self.local_step = 0

    # What the fused routes cache ON the module (scratch pools of update_extra_state -- up to 0.8 GB --, device pointers of the
    # one-call steps, samples marched ahead, events): none of it is model state.  torch.save(model) / copy.deepcopy(model)
    # see the module without it; the copy rebuilds what it needs on first use.
    _TRANSIENT = ("_density_scratch", "_native_ctx", "_native_events_ctx", "_premarched", "_fused_kind",
                  "_pending_density_stats", "_last_march_event")

    def __getstate__(self):
        state = self.__dict__.copy()
        for k in self._TRANSIENT:
            state.pop(k, None)
        return state

    def __deepcopy__(self, memo):
        import copy
        new = self.__class__.__new__(self.__class__)
        memo[id(self)] = new
        for k, v in self.__dict__.items():
            if k not in self._TRANSIENT:
                new.__dict__[k] = copy.deepcopy(v, memo)
        return new

    def reset_extra_state(self):
        if self.cuda_ray:
            for state in (self.density_grid, self.step_counter):
                state.zero_()
            self._zero_counters()

    # what a network provides
    def forward(self, x, d):
        raise NotImplementedError()

    def density(self, x):
        raise NotImplementedError()

    def color(self, x, d, mask=None, **kwargs):
        raise NotImplementedError()

    # ------------------------------------------------------------------------------------------------------ rendering
    def render(self, rays_o, rays_d, staged=False, max_ray_batch=4096, **kwargs):
        """rays [B,N,3] -> {"depth": [B,N], "image": [B,N,C]}.  `staged` bounds the sampler's memory by rendering
        `max_ray_batch` rays at a time; the marching path sizes its own buffers and ignores it."""
        if self.cuda_ray:
            return self.run_cuda(rays_o, rays_d, **kwargs)
        if not staged:
            return self.run(rays_o, rays_d, **kwargs)
        n_img, n_rays = rays_o.shape[0], rays_o.shape[1]
        depth = torch.empty(n_img, n_rays, device=rays_o.device)
        image = torch.empty(n_img, n_rays, self.out_dim_color, device=rays_o.device)
        for b in range(n_img):
            for a in range(0, n_rays, max_ray_batch):
                part = self.run(rays_o[b:b + 1, a:a + max_ray_batch], rays_d[b:b + 1, a:a + max_ray_batch], **kwargs)
                depth[b:b + 1, a:a + max_ray_batch] = part["depth"]
                image[b:b + 1, a:a + max_ray_batch] = part["image"]
        return {"depth": depth, "image": image}

    def run(self, rays_o, rays_d, num_steps=128, upsample_steps=128, bg_color=None, perturb=False, **kwargs):
        return sampler.render_stratified(self, rays_o, rays_d, num_steps, upsample_steps, bg_color, perturb, **kwargs)

    def run_cuda(self, rays_o, rays_d, dt_gamma=0, bg_color=None, perturb=False, force_all_rays=False, max_steps=1024,
                 **kwargs):
        lead = rays_o.shape[:-1]
        rays_o, rays_d = (t.contiguous().view(-1, 3) for t in (rays_o, rays_d))
        if self.bg_radius > 0:
            bg = self.background(raymarching.polar_from_ray(rays_o, rays_d, self.bg_radius), rays_d)
        else:
            bg = 1 if bg_color is None else bg_color
        if self.training:
            if self.bg_radius <= 0 and fused_render.supported(self, rays_o, rays_d, bg, dt_gamma):
                depth, image = fused_render.render_train(self, rays_o, rays_d, bg, perturb, force_all_rays, dt_gamma,
                                                         max_steps)
            else:
                depth, image = self._train_ops(rays_o, rays_d, bg, perturb, force_all_rays, dt_gamma, max_steps)
        elif frame.frame_supported(self, rays_o, perturb, dt_gamma, bg):
            depth, image = frame.render_frame(self, rays_o, rays_d, bg, max_steps)
        else:
            depth, image = frame.render_rounds(self, rays_o, rays_d, bg, perturb, dt_gamma, max_steps)
        return {"depth": depth.view(*lead), "image": image.view(*lead, 3)}

    def _train_ops(self, rays_o, rays_d, bg, perturb, force_all_rays, dt_gamma, max_steps):
        """A training render through the four autograd Functions: slab test -> occupied samples of every
        ray (the step's counts land in this render's row of the `step_counter` ring) -> network -> compositing ->
        background blend; depth is measured from the ray's entry point and normalised by its chord."""
        nears, fars = raymarching.near_far_from_aabb(rays_o, rays_d, self.aabb_train, self.min_near)
        row = self.step_counter[self.local_step % 16]
        row.zero_()
        self.local_step = self.local_step + 1
        xyzs, dirs, deltas, rays = raymarching.march_rays_train(
            rays_o, rays_d, self.bound, self.density_bitfield, self.cascade, self.grid_size, nears, fars, row,
            self.mean_count, perturb, 128, force_all_rays, dt_gamma, max_steps)
        sigmas, rgbs = self(xyzs, dirs)                # (the network's forward: density and colour of every sample)
        weights_sum, depth, image = raymarching.composite_rays_train(self.density_scale * sigmas, rgbs, deltas, rays)
        image = image + (1 - weights_sum).unsqueeze(-1) * bg
        depth = (depth - nears).clamp_(min=0) / (fars - nears)
        return depth, image

    # ---------------------------------------------------------------------------------------- occupancy maintenance
    @torch.no_grad()
    def mark_untrained_grid(self, poses, intrinsic, S=64):
        """Cells no training camera sees get density -1 and are never occupied."""
        if not self.cuda_ray:
            return
        if density_update.supported(self):
            density_update.mark_untrained(self, poses, intrinsic)
        else:
            density_update.mark_untrained_torch(self, poses, intrinsic, cam_chunk=S)

    @property
    def mean_density(self):
        """Mean of the density grid after the last update_extra_state (nerf/renderer.py:547).  The device-side update leaves
        its read-back in flight when the harness already has the sample budget (density_update.update_end(early=...)):
        whoever reads the value -- checkpoints -- gets it resolved here."""
        if getattr(self, "_pending_density_stats", None) is not None:
            density_update.resolve_pending(self)
        return self._mean_density

    @mean_density.setter
    def mean_density(self, value):
        self._mean_density = value

    @torch.no_grad()
    def update_extra_state(self, decay=0.95, S=128):
        """Every 16 training steps: refresh density_grid (EMA-max), density_bitfield, mean_density and the sample budget
        `mean_count`; restart the step_counter ring."""
        if not self.cuda_ray:
            return
        if density_update.supported(self):
            density_update.update(self, decay)
        else:
            density_update.update_torch(self, decay)

    def update_extra_state_begin(self, decay=0.95):
        """update_extra_state with its read-back left open: -> handle for update_extra_state_end, or None when the
        update ran to the end (no device-side pass for this model).  In between the caller may queue whatever does not
        read mean_density / mean_count."""
        if self.cuda_ray and density_update.supported(self):
            return density_update.update_begin(self, decay)
        self.update_extra_state(decay)
        return None

    def update_extra_state_end(self, handle, early=None):
        if handle is not None:
            density_update.update_end(self, handle, early)
