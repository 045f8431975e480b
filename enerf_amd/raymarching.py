"""Host-side mirror of raymarching/raymarching.py (reference): the ten autograd entry points the renderer calls,
with the reference's names, positional orders, defaults and return conventions, on top of the `_raymarching`
backend (HIP).  Differences are confined to what stays invisible to the caller:
  * tensors are moved to the backend's device instead of unconditionally `.cuda()`-ed (the backend is CUDA/HIP
    in the product; tests may inject a CPU oracle backend),
  * `march_rays_train` slot allocation is deterministic (see include/enerf_hip.h).
"""
import torch
from torch.autograd import Function

from .backends import _raymarching as _backend

_DEVICE = "cuda"   # device the backend computes on; tests that inject the CPU oracle backend set this to "cpu"


def _dev(t):
    return t if t.device.type == _DEVICE else t.to(_DEVICE)


def _f32c(t):
    return t.float().contiguous()


# ---------------------------------------------------------------------------- utils
class _near_far_from_aabb(Function):
    @staticmethod
    def forward(ctx, rays_o, rays_d, aabb, min_near=0.2):
        """rays_o/d [N,3], aabb [6] -> nears [N], fars [N]   (raymarching.py:19-49 of the reference)"""
        rays_o = _f32c(_dev(rays_o)).view(-1, 3)
        rays_d = _f32c(_dev(rays_d)).view(-1, 3)
        aabb = _f32c(_dev(aabb))
        N = len(rays_o)
        nears = torch.empty(N, dtype=torch.float32, device=rays_o.device)
        fars = torch.empty(N, dtype=torch.float32, device=rays_o.device)
        _backend.near_far_from_aabb(rays_o, rays_d, aabb, N, min_near, nears, fars)
        return nears, fars


near_far_from_aabb = _near_far_from_aabb.apply


class _polar_from_ray(Function):
    @staticmethod
    def forward(ctx, rays_o, rays_d, radius):
        """rays [N,3] x2 -> (theta, phi) in [-1,1]^2 on the background sphere   (raymarching.py:52-80)"""
        rays_o = _f32c(_dev(rays_o)).view(-1, 3)
        rays_d = _f32c(_dev(rays_d)).view(-1, 3)
        N = len(rays_o)
        coords = torch.empty(N, 2, dtype=torch.float32, device=rays_o.device)
        _backend.polar_from_ray(rays_o, rays_d, radius, N, coords)
        return coords


polar_from_ray = _polar_from_ray.apply


class _morton3D(Function):
    @staticmethod
    def forward(ctx, coords):
        """coords [N,3] int32 in [0,128) -> indices [N] int32   (raymarching.py:83-104)"""
        coords = _dev(coords).int().contiguous()
        N = len(coords)
        indices = coords.new_empty(N)
        _backend.morton3D(coords, N, indices)
        return indices


morton3D = _morton3D.apply


class _morton3D_invert(Function):
    @staticmethod
    def forward(ctx, indices):
        """indices [N] -> coords [N,3] int32   (raymarching.py:106-126)"""
        indices = _dev(indices).int().contiguous()
        N = len(indices)
        coords = indices.new_empty(N, 3)
        _backend.morton3D_invert(indices, N, coords)
        return coords


morton3D_invert = _morton3D_invert.apply


# bumped by everything in this package that writes a density bitfield through a raw pointer (torch's own version counter
# covers copy_ and friends): consumers that cache something derived from a bitfield key it on (data_ptr, _version, this)
BITFIELD_EPOCH = [0]


class _packbits(Function):
    @staticmethod
    def forward(ctx, grid, thresh, bitfield=None):
        """grid [C, H^3] f32 -> bitfield [C*H^3/8] u8, bit i of byte n = grid[8n+i] > thresh   (raymarching.py:129-155)"""
        grid = _f32c(_dev(grid))
        C, H3 = grid.shape[0], grid.shape[1]
        N = C * H3 // 8
        if bitfield is None:
            bitfield = grid.new_empty(N, dtype=torch.uint8)
        _backend.packbits(grid, N, thresh, bitfield)
        BITFIELD_EPOCH[0] += 1
        return bitfield


packbits = _packbits.apply


# ---------------------------------------------------------------------------- training
class _march_rays_train(Function):
    @staticmethod
    def forward(ctx, rays_o, rays_d, bound, density_bitfield, C, H, nears, fars, step_counter=None, mean_count=-1,
                perturb=False, align=-1, force_all_rays=False, dt_gamma=0, max_steps=1024):
        """Generate the occupied samples of every ray (forward only).   (raymarching.py:161-230)

        Returns xyzs [M,3], dirs [M,3], deltas [M,2] (dt, real delta-t), rays [N,3] int32 (ray id, offset, count).
        M = N*max_steps while `mean_count <= 0` (then cropped to the used count rounded up by `align`, which costs a
        device->host read of step_counter[0]); afterwards M = mean_count rounded *up past* the next multiple of
        `align` and rays that do not fit are dropped (their rows stay zero).
        """
        rays_o = _f32c(_dev(rays_o)).view(-1, 3)
        rays_d = _f32c(_dev(rays_d)).view(-1, 3)
        density_bitfield = _dev(density_bitfield).contiguous()
        nears = _f32c(_dev(nears))
        fars = _f32c(_dev(fars))

        N = len(rays_o)
        M = N * max_steps
        if not force_all_rays and mean_count > 0:
            if align > 0:
                mean_count += align - mean_count % align
            M = mean_count

        dev = rays_o.device
        xyzs = torch.zeros(M, 3, dtype=torch.float32, device=dev)
        dirs = torch.zeros(M, 3, dtype=torch.float32, device=dev)
        deltas = torch.zeros(M, 2, dtype=torch.float32, device=dev)
        rays = torch.empty(N, 3, dtype=torch.int32, device=dev)
        if step_counter is None:
            step_counter = torch.zeros(2, dtype=torch.int32, device=dev)

        _backend.march_rays_train(rays_o, rays_d, density_bitfield, bound, dt_gamma, max_steps, N, C, H, M, nears,
                                  fars, xyzs, dirs, deltas, rays, step_counter, perturb)

        if force_all_rays or mean_count <= 0:
            m = step_counter[0].item()
            if align > 0:
                m += align - m % align
            xyzs, dirs, deltas = xyzs[:m], dirs[:m], deltas[:m]
        return xyzs, dirs, deltas, rays


march_rays_train = _march_rays_train.apply


class _composite_rays_train(Function):
    @staticmethod
    def forward(ctx, sigmas, rgbs, deltas, rays):
        """Front-to-back compositing of every ray's samples.   (raymarching.py:233-264)
        sigmas [M], rgbs [M,3], deltas [M,2], rays [N,3] -> weights_sum [N], depth [N], image [N,3]"""
        sigmas = _f32c(sigmas)
        rgbs = _f32c(rgbs)
        deltas = _f32c(deltas)
        M, N = sigmas.shape[0], rays.shape[0]
        dev = sigmas.device
        weights_sum = torch.empty(N, dtype=torch.float32, device=dev)
        depth = torch.empty(N, dtype=torch.float32, device=dev)
        image = torch.empty(N, 3, dtype=torch.float32, device=dev)
        _backend.composite_rays_train_forward(sigmas, rgbs, deltas, rays, M, N, weights_sum, depth, image)
        ctx.save_for_backward(sigmas, rgbs, deltas, rays, weights_sum, depth, image)
        ctx.dims = (M, N)
        return weights_sum, depth, image

    @staticmethod
    def backward(ctx, grad_weights_sum, grad_depth, grad_image):
        # grad_depth is not propagated (as in the reference, raymarching.py:270)
        sigmas, rgbs, deltas, rays, weights_sum, depth, image = ctx.saved_tensors
        M, N = ctx.dims
        grad_weights_sum = _f32c(grad_weights_sum)
        grad_image = _f32c(grad_image)
        grad_sigmas, grad_rgbs = torch.zeros_like(sigmas), torch.zeros_like(rgbs)
        _backend.composite_rays_train_backward(grad_weights_sum, grad_image, sigmas, rgbs, deltas, rays, weights_sum,
                                               image, M, N, grad_sigmas, grad_rgbs)
        return grad_sigmas, grad_rgbs, None, None


composite_rays_train = _composite_rays_train.apply


# ---------------------------------------------------------------------------- inference
class _march_rays(Function):
    @staticmethod
    def forward(ctx, n_alive, n_step, rays_alive, rays_t, rays_o, rays_d, bound, density_bitfield, C, H, near, far,
                align=-1, perturb=False, dt_gamma=0, max_steps=1024):
        """March `n_step` occupied samples for each alive ray.   (raymarching.py:292-337)
        Returns xyzs/dirs [n_alive*n_step (+pad), 3] and deltas [.., 2]; unused slots are zero (delta == 0 marks end)."""
        rays_o = _f32c(_dev(rays_o)).view(-1, 3)
        rays_d = _f32c(_dev(rays_d)).view(-1, 3)
        M = n_alive * n_step
        if align > 0:
            M += align - (M % align)
        dev = rays_o.device
        if hasattr(_backend, "march_rays_ex"):
            # the HIP marcher zero-fills what it does not write: no torch.zeros passes over the sample buffers
            xyzs = torch.empty(M, 3, dtype=torch.float32, device=dev)
            dirs = torch.empty(M, 3, dtype=torch.float32, device=dev)
            deltas = torch.empty(M, 2, dtype=torch.float32, device=dev)
            _backend.march_rays_ex(n_alive, n_step, rays_alive, rays_t, rays_o, rays_d, bound, dt_gamma, max_steps, C,
                                   H, density_bitfield, near, far, xyzs, dirs, deltas, perturb)
            return xyzs, dirs, deltas
        xyzs = torch.zeros(M, 3, dtype=torch.float32, device=dev)
        dirs = torch.zeros(M, 3, dtype=torch.float32, device=dev)
        deltas = torch.zeros(M, 2, dtype=torch.float32, device=dev)
        _backend.march_rays(n_alive, n_step, rays_alive, rays_t, rays_o, rays_d, bound, dt_gamma, max_steps, C, H,
                            density_bitfield, near, far, xyzs, dirs, deltas, perturb)
        return xyzs, dirs, deltas


march_rays = _march_rays.apply


class _composite_rays(Function):
    @staticmethod
    def forward(ctx, n_alive, n_step, rays_alive, rays_t, sigmas, rgbs, deltas, weights_sum, depth, image):
        """In-place accumulation into weights_sum / depth / image; rays_t[n] = -1 marks a finished ray.
        (raymarching.py:340-362)"""
        _backend.composite_rays(n_alive, n_step, rays_alive, rays_t, _f32c(sigmas), _f32c(rgbs), deltas, weights_sum,
                                depth, image)
        return tuple()


composite_rays = _composite_rays.apply


class _compact_rays(Function):
    @staticmethod
    def forward(ctx, n_alive, rays_alive, rays_alive_old, rays_t, rays_t_old, alive_counter):
        """Keep rays with rays_t_old >= 0 (stable order); alive_counter[0] += survivors.   (raymarching.py:365-382)"""
        _backend.compact_rays(n_alive, rays_alive, rays_alive_old, rays_t, rays_t_old, alive_counter)
        return tuple()


compact_rays = _compact_rays.apply
