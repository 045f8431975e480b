"""`NeRFRenderer` -- the caller of the hot path, re-stated from nerf/renderer.py of the reference so that the bench
harness and the tests can drive the kernels exactly the way E-NeRF does.

Same constructor arguments, buffer names / shapes (`aabb_train`, `aabb_infer`, `density_grid [cas,128^3]`,
`density_bitfield [cas*128^3/8] u8`, `step_counter [16,2] i32` -- state_dict compatible), same keyword contract for
`render` / `run` / `run_cuda`, same sequencing of the native calls:

    run_cuda (train):  near_far_from_aabb -> march_rays_train -> self(xyzs, dirs) -> composite_rays_train
    run_cuda (eval):   near_far_from_aabb -> loop[compact_rays -> march_rays -> self(...) -> composite_rays]
    run:               near_far_from_aabb -> stratified samples -> density() -> cumprod compositing -> color()
    update_extra_state: morton3D(/invert) -> density() -> EMA max -> packbits

Reference lines are cited per method.
"""
import math

import numpy as np
import torch
import torch.nn as nn

from . import density_update, fused_network, fused_render, raymarching


def _meshgrid_ij(*args):
    return torch.meshgrid(*args, indexing="ij")


def sample_pdf(bins, weights, n_samples, det=False):
    """Inverse-CDF resampling (nerf/renderer.py:12-46).  bins [B,T], weights [B,T-1] -> [B,n_samples]."""
    weights = weights + 1e-5
    pdf = weights / torch.sum(weights, -1, keepdim=True)
    cdf = torch.cumsum(pdf, -1)
    cdf = torch.cat([torch.zeros_like(cdf[..., :1]), cdf], -1)
    if det:
        u = torch.linspace(0.0 + 0.5 / n_samples, 1.0 - 0.5 / n_samples, steps=n_samples).to(weights.device)
        u = u.expand(list(cdf.shape[:-1]) + [n_samples])
    else:
        u = torch.rand(list(cdf.shape[:-1]) + [n_samples]).to(weights.device)
    u = u.contiguous()
    inds = torch.searchsorted(cdf, u, right=True)
    below = torch.clamp(inds - 1, min=0)
    above = torch.clamp(inds, max=cdf.shape[-1] - 1)
    inds_g = torch.stack([below, above], -1)
    shape = [inds_g.shape[0], inds_g.shape[1], cdf.shape[-1]]
    cdf_g = torch.gather(cdf.unsqueeze(1).expand(shape), 2, inds_g)
    bins_g = torch.gather(bins.unsqueeze(1).expand(shape), 2, inds_g)
    denom = cdf_g[..., 1] - cdf_g[..., 0]
    denom = torch.where(denom < 1e-5, torch.ones_like(denom), denom)
    t = (u - cdf_g[..., 0]) / denom
    return bins_g[..., 0] + t * (bins_g[..., 1] - bins_g[..., 0])


class NeRFRenderer(nn.Module):
    def __init__(self, bound=1, cuda_ray=False, density_scale=1, min_near=0.2, density_thresh=0.01, bg_radius=-1):
        super().__init__()
        self.bound = bound
        self.cascade = 1 + math.ceil(math.log2(bound))
        self.grid_size = 128
        self.density_scale = density_scale
        self.min_near = min_near
        self.density_thresh = density_thresh
        self.bg_radius = bg_radius

        aabb_train = torch.FloatTensor([-bound, -bound, -bound, bound, bound, bound])
        self.register_buffer("aabb_train", aabb_train)
        self.register_buffer("aabb_infer", aabb_train.clone())

        self.cuda_ray = cuda_ray
        if cuda_ray:
            self.register_buffer("density_grid", torch.zeros([self.cascade, self.grid_size ** 3]))
            self.register_buffer("density_bitfield",
                                 torch.zeros(self.cascade * self.grid_size ** 3 // 8, dtype=torch.uint8))
            self.mean_density = 0
            self.iter_density = 0
            self.register_buffer("step_counter", torch.zeros(16, 2, dtype=torch.int32))
            self.mean_count = 0
            self.local_step = 0

    def forward(self, x, d):
        raise NotImplementedError()

    def density(self, x):
        raise NotImplementedError()

    def color(self, x, d, mask=None, **kwargs):
        raise NotImplementedError()

    def reset_extra_state(self):
        if not self.cuda_ray:
            return
        self.density_grid.zero_()
        self.mean_density = 0
        self.iter_density = 0
        self.step_counter.zero_()
        self.mean_count = 0
        self.local_step = 0

    # ------------------------------------------------------------------ nerf/renderer.py:150-278
    def run(self, rays_o, rays_d, num_steps=128, upsample_steps=128, bg_color=None, perturb=False, **kwargs):
        prefix = rays_o.shape[:-1]
        rays_o = rays_o.contiguous().view(-1, 3)
        rays_d = rays_d.contiguous().view(-1, 3)
        N = rays_o.shape[0]
        device = rays_o.device
        aabb = self.aabb_train if self.training else self.aabb_infer

        nears, fars = raymarching.near_far_from_aabb(rays_o, rays_d, aabb, self.min_near)
        nears = nears.to(device).unsqueeze(-1)
        fars = fars.to(device).unsqueeze(-1)

        z_vals = torch.linspace(0.0, 1.0, num_steps, device=device).unsqueeze(0).expand((N, num_steps))
        z_vals = nears + (fars - nears) * z_vals
        sample_dist = (fars - nears) / num_steps
        if perturb:
            z_vals = z_vals + (torch.rand(z_vals.shape, device=device) - 0.5) * sample_dist

        xyzs = rays_o.unsqueeze(-2) + rays_d.unsqueeze(-2) * z_vals.unsqueeze(-1)
        xyzs = torch.min(torch.max(xyzs, aabb[:3]), aabb[3:])

        density_outputs = self.density(xyzs.reshape(-1, 3))
        for k, v in density_outputs.items():
            density_outputs[k] = v.view(N, num_steps, -1)

        if upsample_steps > 0:
            with torch.no_grad():
                deltas = z_vals[..., 1:] - z_vals[..., :-1]
                deltas = torch.cat([deltas, sample_dist * torch.ones_like(deltas[..., :1])], dim=-1)
                alphas = 1 - torch.exp(-deltas * self.density_scale * density_outputs["sigma"].squeeze(-1))
                alphas_shifted = torch.cat([torch.ones_like(alphas[..., :1]), 1 - alphas + 1e-15], dim=-1)
                weights = alphas * torch.cumprod(alphas_shifted, dim=-1)[..., :-1]
                z_vals_mid = z_vals[..., :-1] + 0.5 * deltas[..., :-1]
                new_z_vals = sample_pdf(z_vals_mid, weights[:, 1:-1], upsample_steps, det=not self.training).detach()
                new_xyzs = rays_o.unsqueeze(-2) + rays_d.unsqueeze(-2) * new_z_vals.unsqueeze(-1)
                new_xyzs = torch.min(torch.max(new_xyzs, aabb[:3]), aabb[3:])
            new_density_outputs = self.density(new_xyzs.reshape(-1, 3))
            for k, v in new_density_outputs.items():
                new_density_outputs[k] = v.view(N, upsample_steps, -1)
            z_vals = torch.cat([z_vals, new_z_vals], dim=1)
            z_vals, z_index = torch.sort(z_vals, dim=1)
            xyzs = torch.cat([xyzs, new_xyzs], dim=1)
            xyzs = torch.gather(xyzs, dim=1, index=z_index.unsqueeze(-1).expand_as(xyzs))
            for k in density_outputs:
                tmp = torch.cat([density_outputs[k], new_density_outputs[k]], dim=1)
                density_outputs[k] = torch.gather(tmp, dim=1, index=z_index.unsqueeze(-1).expand_as(tmp))

        deltas = z_vals[..., 1:] - z_vals[..., :-1]
        deltas = torch.cat([deltas, sample_dist * torch.ones_like(deltas[..., :1])], dim=-1)
        alphas = 1 - torch.exp(-deltas * self.density_scale * density_outputs["sigma"].squeeze(-1))
        alphas_shifted = torch.cat([torch.ones_like(alphas[..., :1]), 1 - alphas + 1e-15], dim=-1)
        weights = alphas * torch.cumprod(alphas_shifted, dim=-1)[..., :-1]

        mask = weights > 1e-4
        dirs = rays_d.view(-1, 1, 3).expand_as(xyzs)
        for k, v in density_outputs.items():
            density_outputs[k] = v.view(-1, v.shape[-1])
        rgbs = self.color(xyzs.reshape(-1, 3), dirs.reshape(-1, 3), mask=mask.reshape(-1), **density_outputs)
        rgbs = rgbs.view(N, -1, kwargs["out_dim_color"])

        weights_sum = weights.sum(dim=-1)
        ori_z_vals = ((z_vals - nears) / (fars - nears)).clamp(0, 1)
        depth = torch.sum(weights * ori_z_vals, dim=-1)
        image = torch.sum(weights.unsqueeze(-1) * rgbs, dim=-2)

        if self.bg_radius > 0:
            polar = raymarching.polar_from_ray(rays_o, rays_d, self.bg_radius)
            bg_color = self.background(polar, rays_d.reshape(-1, 3))
        elif bg_color is None:
            bg_color = 1
        image = image + (1 - weights_sum).unsqueeze(-1) * bg_color
        image = image.view(*prefix, kwargs["out_dim_color"])
        depth = depth.view(*prefix)
        return {"depth": depth, "image": image}

    # ------------------------------------------------------------------ nerf/renderer.py:281-406
    def run_cuda(self, rays_o, rays_d, dt_gamma=0, bg_color=None, perturb=False, force_all_rays=False, max_steps=1024,
                 **kwargs):
        prefix = rays_o.shape[:-1]
        rays_o = rays_o.contiguous().view(-1, 3)
        rays_d = rays_d.contiguous().view(-1, 3)
        N = rays_o.shape[0]
        device = rays_o.device

        if self.training and bg_color is not None and fused_render.supported(self, rays_o, rays_d, bg_color, dt_gamma):
            depth, image = fused_render.render_train(self, rays_o, rays_d, bg_color, perturb, force_all_rays, dt_gamma,
                                                     max_steps)
            return {"depth": depth.view(*prefix), "image": image.view(*prefix, 3)}
        if self.training and bg_color is None and self.bg_radius <= 0 \
                and fused_render.supported(self, rays_o, rays_d, 1, dt_gamma):
            depth, image = fused_render.render_train(self, rays_o, rays_d, 1, perturb, force_all_rays, dt_gamma,
                                                     max_steps)
            return {"depth": depth.view(*prefix), "image": image.view(*prefix, 3)}

        nears, fars = raymarching.near_far_from_aabb(rays_o, rays_d,
                                                     self.aabb_train if self.training else self.aabb_infer,
                                                     self.min_near)
        if self.bg_radius > 0:
            polar = raymarching.polar_from_ray(rays_o, rays_d, self.bg_radius)
            bg_color = self.background(polar, rays_d)
        elif bg_color is None:
            bg_color = 1

        if self.training:
            counter = self.step_counter[self.local_step % 16]
            counter.zero_()
            self.local_step += 1

            xyzs, dirs, deltas, rays = raymarching.march_rays_train(
                rays_o, rays_d, self.bound, self.density_bitfield, self.cascade, self.grid_size, nears, fars, counter,
                self.mean_count, perturb, 128, force_all_rays, dt_gamma, max_steps)
            sigmas, rgbs = self(xyzs, dirs)
            sigmas = self.density_scale * sigmas
            weights_sum, depth, image = raymarching.composite_rays_train(sigmas, rgbs, deltas, rays)
            image = image + (1 - weights_sum).unsqueeze(-1) * bg_color
            depth = torch.clamp(depth - nears, min=0) / (fars - nears)
            image = image.view(*prefix, 3)
            depth = depth.view(*prefix)
        else:
            dtype = torch.float32
            weights_sum = torch.zeros(N, dtype=dtype, device=device)
            depth = torch.zeros(N, dtype=dtype, device=device)
            image = torch.zeros(N, 3, dtype=dtype, device=device)

            n_alive = N
            alive_counter = torch.zeros([1], dtype=torch.int32, device=device)
            rays_alive = torch.zeros(2, n_alive, dtype=torch.int32, device=device)
            rays_t = torch.zeros(2, n_alive, dtype=dtype, device=device)

            step = 0
            i = 0
            while step < 1024:
                if step == 0:
                    torch.arange(n_alive, out=rays_alive[0])
                    rays_t[0] = nears
                else:
                    alive_counter.zero_()
                    raymarching.compact_rays(n_alive, rays_alive[i % 2], rays_alive[(i + 1) % 2], rays_t[i % 2],
                                             rays_t[(i + 1) % 2], alive_counter)
                    n_alive = alive_counter.item()
                if n_alive <= 0:
                    break
                # reference schedule (infer_batch_mult = 1): n_alive * n_step <= N samples per iteration, n_step <= 8.
                # A 288 GB part can take K times more samples per iteration; the per-ray sample sequence and the
                # compositing order do not depend on the chunking, so the image is bit-identical, with ~K x fewer
                # iterations (each of which costs a device->host read of the alive counter).
                K = max(int(getattr(self, "infer_batch_mult", 1)), 1)
                n_step = max(min(K * N // n_alive, 8 * K), 1)
                xyzs, dirs, deltas = raymarching.march_rays(n_alive, n_step, rays_alive[i % 2], rays_t[i % 2], rays_o,
                                                            rays_d, self.bound, self.density_bitfield, self.cascade,
                                                            self.grid_size, nears, fars, 128, perturb, dt_gamma,
                                                            max_steps)
                sigmas, rgbs = self(xyzs, dirs)
                sigmas = self.density_scale * sigmas
                raymarching.composite_rays(n_alive, n_step, rays_alive[i % 2], rays_t[i % 2], sigmas, rgbs, deltas,
                                           weights_sum, depth, image)
                step += n_step
                i += 1

            image = image + (1 - weights_sum).unsqueeze(-1) * bg_color
            depth = torch.clamp(depth - nears, min=0) / (fars - nears)
            image = image.view(*prefix, 3)
            depth = depth.view(*prefix)
        return {"depth": depth, "image": image}

    # ------------------------------------------------------------------ nerf/renderer.py:408-469
    @torch.no_grad()
    def mark_untrained_grid(self, poses, intrinsic, S=64):
        if not self.cuda_ray:
            return
        if density_update.supported(self):
            return density_update.mark_untrained(self, poses, intrinsic)       # one launch instead of the 5-level loop
        if isinstance(poses, np.ndarray):
            poses = torch.from_numpy(poses)
        B = poses.shape[0]
        fx, fy, cx, cy = intrinsic
        dev = self.density_grid.device
        X = torch.arange(self.grid_size, dtype=torch.int32, device=dev).split(S)
        Y = torch.arange(self.grid_size, dtype=torch.int32, device=dev).split(S)
        Z = torch.arange(self.grid_size, dtype=torch.int32, device=dev).split(S)
        count = torch.zeros_like(self.density_grid)
        poses = poses.to(dev)
        for xs in X:
            for ys in Y:
                for zs in Z:
                    xx, yy, zz = _meshgrid_ij(xs, ys, zs)
                    coords = torch.cat([xx.reshape(-1, 1), yy.reshape(-1, 1), zz.reshape(-1, 1)], dim=-1)
                    indices = raymarching.morton3D(coords).long()
                    world_xyzs = (2 * coords.float() / (self.grid_size - 1) - 1).unsqueeze(0)
                    for cas in range(self.cascade):
                        bound = min(2 ** cas, self.bound)
                        half_grid_size = bound / self.grid_size
                        cas_world_xyzs = world_xyzs * (bound - half_grid_size)
                        head = 0
                        while head < B:
                            tail = min(head + S, B)
                            cam_xyzs = cas_world_xyzs - poses[head:tail, :3, 3].unsqueeze(1)
                            cam_xyzs = cam_xyzs @ poses[head:tail, :3, :3]
                            mask_z = cam_xyzs[:, :, 2] > 0
                            mask_x = torch.abs(cam_xyzs[:, :, 0]) < cx / fx * cam_xyzs[:, :, 2] + half_grid_size * 2
                            mask_y = torch.abs(cam_xyzs[:, :, 1]) < cy / fy * cam_xyzs[:, :, 2] + half_grid_size * 2
                            mask = (mask_z & mask_x & mask_y).sum(0).reshape(-1)
                            count[cas, indices] += mask
                            head += S
        self.density_grid[count == 0] = -1

    # ------------------------------------------------------------------ nerf/renderer.py:473-561
    @torch.no_grad()
    def update_extra_state(self, decay=0.95, S=128):
        if not self.cuda_ray:
            return
        if density_update.supported(self):
            return density_update.update(self, decay)        # device-side selection / EMA / packbits, one read-back
        dev = self.density_grid.device
        tmp_grid = -torch.ones_like(self.density_grid)

        if self.iter_density < 16:   # full update
            X = torch.arange(self.grid_size, dtype=torch.int32, device=dev).split(S)
            Y = torch.arange(self.grid_size, dtype=torch.int32, device=dev).split(S)
            Z = torch.arange(self.grid_size, dtype=torch.int32, device=dev).split(S)
            for xs in X:
                for ys in Y:
                    for zs in Z:
                        # same cell set as the reference's meshgrid(xs, ys, zs), enumerated with x fastest: consecutive
                        # query points are then x-neighbours, i.e. neighbouring rows of every level of the hash grid
                        # (dense levels index x + y*R + z*R^2; hashed levels x ^ const) -> coalesced gathers
                        zz, yy, xx = _meshgrid_ij(zs, ys, xs)
                        coords = torch.cat([xx.reshape(-1, 1), yy.reshape(-1, 1), zz.reshape(-1, 1)], dim=-1)
                        indices = raymarching.morton3D(coords).long()
                        xyzs = 2 * coords.float() / (self.grid_size - 1) - 1
                        for cas in range(self.cascade):
                            tmp_grid[cas, indices] = self._cell_density(xyzs, cas)
        else:                        # partial update: N uniform cells + N occupied cells per cascade
            N = self.grid_size ** 3 // 4
            for cas in range(self.cascade):
                coords = torch.randint(0, self.grid_size, (N, 3), device=dev)
                indices = raymarching.morton3D(coords).long()
                occ_indices = torch.nonzero(self.density_grid[cas] > 0).squeeze(-1)
                rand_mask = torch.randint(0, occ_indices.shape[0], [N], dtype=torch.long, device=dev)
                occ_indices = occ_indices[rand_mask]
                indices = torch.cat([indices, occ_indices], dim=0)
                # spatially sorted (morton) evaluation order: same cells, cache-friendly gathers
                indices = torch.sort(indices)[0]
                coords = raymarching.morton3D_invert(indices)
                xyzs = 2 * coords.float() / (self.grid_size - 1) - 1
                tmp_grid[cas, indices] = self._cell_density(xyzs, cas)

        valid_mask = (self.density_grid >= 0) & (tmp_grid >= 0)
        self.density_grid[valid_mask] = torch.maximum(self.density_grid[valid_mask] * decay, tmp_grid[valid_mask])
        self.mean_density = torch.mean(self.density_grid.clamp(min=0)).item()
        self.iter_density += 1

        density_thresh = min(self.mean_density, self.density_thresh)
        self.density_bitfield = raymarching.packbits(self.density_grid, density_thresh, self.density_bitfield)

        total_step = min(16, self.local_step)
        if total_step > 0:
            self.mean_count = int(self.step_counter[:total_step, 0].sum().item() / total_step)
        self.local_step = 0

    def _cell_density(self, xyzs, cas):
        """Jittered cell-centre density of cascade `cas`, scaled by density_scale * dt_min (renderer.py:499-513)."""
        bound = min(2 ** cas, self.bound)
        half_grid_size = bound / self.grid_size
        cas_xyzs = xyzs * (bound - half_grid_size)
        cas_xyzs += (torch.rand_like(cas_xyzs) * 2 - 1) * half_grid_size
        if cas_xyzs.dim() == 2 and fused_network.supported(self, cas_xyzs, cas_xyzs):
            sigmas = fused_network.density_sigma(self, cas_xyzs)       # sigma only: no geo_feat written
        else:
            sigmas = self.density(cas_xyzs)["sigma"].reshape(-1).detach()
        sigmas *= self.density_scale * 0.003383
        return sigmas

    # ------------------------------------------------------------------ nerf/renderer.py:566-599
    def render(self, rays_o, rays_d, staged=False, max_ray_batch=4096, **kwargs):
        _run = self.run_cuda if self.cuda_ray else self.run
        B, N = rays_o.shape[:2]
        device = rays_o.device
        if staged and not self.cuda_ray:
            depth = torch.empty((B, N), device=device)
            image = torch.empty((B, N, self.out_dim_color), device=device)
            for b in range(B):
                head = 0
                while head < N:
                    tail = min(head + max_ray_batch, N)
                    results_ = _run(rays_o[b:b + 1, head:tail], rays_d[b:b + 1, head:tail], **kwargs)
                    depth[b:b + 1, head:tail] = results_["depth"]
                    image[b:b + 1, head:tail] = results_["image"]
                    head += max_ray_batch
            return {"depth": depth, "image": image}
        return _run(rays_o, rays_d, **kwargs)
