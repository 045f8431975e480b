"""The pure-PyTorch sampler of the renderer: `NeRFRenderer.run` (reference nerf/renderer.py:150-278).

This is the route every shipped config of the reference takes when `cuda_ray` is off, and the one bench.py times on
the host cores as `cpu_baseline`.  It is written as four small pieces so that the tests can pin each against the
reference-minted fixtures (tests/golden/ref_run_*.npz) and against the native compositing kernels:

    stratified_depths    T depths per ray, evenly spaced in [near, far], optionally jittered by half a bin
    ray_weights          opacity of every sample -> compositing weight (exclusive transmittance product)
    resample_depths      inverse-CDF draw of extra depths where the coarse weights are large
    render_stratified    the whole render: depths -> density -> [resample -> density -> merge] -> colour -> pixel

Nothing here is on the MI355X hot path (that is run_cuda: renderer.py / fused_render.py / frame.py).
"""
import torch

from . import raymarching


def stratified_depths(nears, fars, n, jitter):
    """nears, fars [N,1] -> depths [N,n] and the bin width [N,1] (`(far - near) / n`, the reference's sample_dist)."""
    grid = torch.linspace(0.0, 1.0, n, device=nears.device)
    span = fars - nears
    z = nears + span * grid.unsqueeze(0).expand(nears.shape[0], n)
    width = span / n
    if jitter:
        z = z + (torch.rand(z.shape, device=nears.device) - 0.5) * width
    return z, width


def ray_weights(z, sigma, tail, density_scale):
    """Compositing weight of every sample of every ray.

    z [N,T] sorted depths, sigma [N,T], tail [N,1] = interval assigned to the last sample.
    alpha_k = 1 - exp(-(z_{k+1} - z_k) * density_scale * sigma_k);  w_k = alpha_k * prod_{j<k} (1 - alpha_j + 1e-15)."""
    step = torch.cat([z[:, 1:] - z[:, :-1], tail * torch.ones_like(z[:, :1])], dim=-1)
    alpha = 1 - torch.exp(-step * density_scale * sigma)
    through = torch.cumprod(torch.cat([torch.ones_like(alpha[:, :1]), 1 - alpha + 1e-15], dim=-1), dim=-1)
    return alpha * through[:, :-1], step


def resample_depths(edges, w, n, deterministic):
    """Inverse-CDF sampling: edges [N,B] (bin boundaries), w [N,B-1] (weight per bin) -> n new depths per ray [N,n].
    The piecewise-constant pdf is w + 1e-5, normalised; u is a regular comb (evaluation) or uniform draws (training:
    drawn on the host generator and moved, like the reference, so a seeded run consumes the same stream)."""
    pdf = w + 1e-5
    pdf = pdf / pdf.sum(-1, keepdim=True)
    cdf = torch.cat([torch.zeros_like(pdf[:, :1]), torch.cumsum(pdf, -1)], dim=-1)                # [N,B]
    N = cdf.shape[0]
    if deterministic:
        u = torch.linspace(0.5 / n, 1.0 - 0.5 / n, steps=n).to(w.device).expand(N, n)
    else:
        u = torch.rand(N, n).to(w.device)
    u = u.contiguous()
    hi = torch.searchsorted(cdf, u, right=True)
    lo = (hi - 1).clamp(min=0)
    hi = hi.clamp(max=cdf.shape[-1] - 1)
    c_lo, c_hi = torch.gather(cdf, 1, lo), torch.gather(cdf, 1, hi)
    e_lo, e_hi = torch.gather(edges, 1, lo), torch.gather(edges, 1, hi)
    mass = c_hi - c_lo
    mass = torch.where(mass < 1e-5, torch.ones_like(mass), mass)
    return e_lo + (u - c_lo) / mass * (e_hi - e_lo)


def _points(rays_o, rays_d, z, aabb):
    p = rays_o.unsqueeze(1) + rays_d.unsqueeze(1) * z.unsqueeze(-1)
    return torch.min(torch.max(p, aabb[:3]), aabb[3:])


def render_stratified(model, rays_o, rays_d, num_steps=128, upsample_steps=128, bg_color=None, perturb=False, **kwargs):
    """-> {"depth": [...], "image": [..., out_dim_color]} for rays of any leading shape."""
    lead = rays_o.shape[:-1]
    rays_o = rays_o.contiguous().view(-1, 3)
    rays_d = rays_d.contiguous().view(-1, 3)
    N, dev = rays_o.shape[0], rays_o.device
    out_c = kwargs["out_dim_color"]
    aabb = model.aabb_train if model.training else model.aabb_infer

    nears, fars = raymarching.near_far_from_aabb(rays_o, rays_d, aabb, model.min_near)
    nears, fars = nears.to(dev).unsqueeze(-1), fars.to(dev).unsqueeze(-1)
    z, width = stratified_depths(nears, fars, num_steps, perturb)
    pts = _points(rays_o, rays_d, z, aabb)
    fields = {k: v.view(N, num_steps, -1) for k, v in model.density(pts.reshape(-1, 3)).items()}

    if upsample_steps > 0:
        with torch.no_grad():
            w, step = ray_weights(z, fields["sigma"].squeeze(-1), width, model.density_scale)
            mids = z[:, :-1] + 0.5 * step[:, :-1]
            z_new = resample_depths(mids, w[:, 1:-1], upsample_steps, deterministic=not model.training).detach()
            pts_new = _points(rays_o, rays_d, z_new, aabb)
        fields_new = {k: v.view(N, upsample_steps, -1) for k, v in model.density(pts_new.reshape(-1, 3)).items()}
        # merge coarse and fine samples in depth order
        z, order = torch.sort(torch.cat([z, z_new], dim=1), dim=1)
        pick = lambda a, b: torch.gather(torch.cat([a, b], dim=1), 1, order.unsqueeze(-1).expand(-1, -1, a.shape[-1]))  # noqa: E731
        pts = pick(pts, pts_new)
        fields = {k: pick(fields[k], fields_new[k]) for k in fields}

    w, _ = ray_weights(z, fields["sigma"].squeeze(-1), width, model.density_scale)
    flat = {k: v.reshape(-1, v.shape[-1]) for k, v in fields.items()}
    dirs = rays_d.view(N, 1, 3).expand_as(pts)
    rgb = model.color(pts.reshape(-1, 3), dirs.reshape(-1, 3), mask=(w > 1e-4).reshape(-1), **flat).view(N, -1, out_c)

    opacity = w.sum(-1)
    depth = (w * ((z - nears) / (fars - nears)).clamp(0, 1)).sum(-1)
    image = (w.unsqueeze(-1) * rgb).sum(-2)
    if model.bg_radius > 0:
        bg_color = model.background(raymarching.polar_from_ray(rays_o, rays_d, model.bg_radius), rays_d.reshape(-1, 3))
    elif bg_color is None:
        bg_color = 1
    image = image + (1 - opacity).unsqueeze(-1) * bg_color
    return {"depth": depth.view(*lead), "image": image.view(*lead, out_c)}
