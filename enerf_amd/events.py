"""The event-loss side of the hot path's caller, re-stated from the reference:

  get_rays / get_event_rays     nerf/utils.py:110-174, 184-216
  rgb_to_luma / lin_log         utils/event_utils.py:23-66
  event_loss / train_step_events   nerf/utils.py:482-573  (Trainer.train_step_events)

`train_step_events` takes the model plus a plain options object instead of a Trainer, but renders, converts and
penalises exactly as the reference does (two renders sharing one random background colour; (lin-)log intensity
difference against polarity * C_thres, or the normalised variant when C_thres == -1).
"""
import numpy as np
import torch


def get_rays(poses, intrinsics, H, W, N=-1, inds=None):
    """poses [B,4,4] cam2world, intrinsics (fx,fy,cx,cy) -> rays_o, rays_d [B,N,3] (+ inds [B,N] when sampled).
    Pixel centres are integer coordinates (no +0.5), as in the reference."""
    device = poses.device
    B = poses.shape[0]
    fx, fy, cx, cy = intrinsics
    i, j = torch.meshgrid(torch.linspace(0, W - 1, W, device=device), torch.linspace(0, H - 1, H, device=device),
                          indexing="ij")
    i = i.t().reshape([1, H * W]).expand([B, H * W])
    j = j.t().reshape([1, H * W]).expand([B, H * W])
    results = {}
    if N > 0:
        N = min(N, H * W)
        if inds is None:
            inds = torch.randint(0, H * W, size=[N], device=device)
        inds = inds.expand([B, N])
        i = torch.gather(i, -1, inds)
        j = torch.gather(j, -1, inds)
        results["inds"] = inds
    zs = torch.ones_like(i)
    xs = (i - cx) / fx * zs
    ys = (j - cy) / fy * zs
    directions = torch.stack((xs, ys, zs), dim=-1)
    directions = directions / torch.norm(directions, dim=-1, keepdim=True)
    rays_d = directions @ poses[:, :3, :3].transpose(-1, -2)
    rays_o = poses[..., :3, 3][..., None, :].expand_as(rays_d)
    results["rays_o"] = rays_o
    results["rays_d"] = rays_d
    return results


def get_event_rays(xs, ys, c2w_before, c2w_at, intrinsics):
    """Per-event ray pairs: pixel (xs, ys) seen from the pose just before and at the event. c2w_* [B,Nevs,3,4]."""
    fx, fy, cx, cy = intrinsics
    zs = torch.ones_like(xs)
    us = (xs - cx) / fx * zs
    vs = (ys - cy) / fy * zs
    dirs_cams = torch.stack((us, vs, zs), dim=-1)
    dirs_cams = dirs_cams / torch.norm(dirs_cams, dim=-1, keepdim=True)
    return {
        "rays_evs_o1": c2w_before[..., :3, 3],
        "rays_evs_d1": torch.sum(dirs_cams[..., None, :] * c2w_before[..., :3, :3], axis=-1),
        "rays_evs_o2": c2w_at[..., :3, 3],
        "rays_evs_d2": torch.sum(dirs_cams[..., None, :] * c2w_at[..., :3, :3], axis=-1),
    }


_luma_factors = {}


def rgb_to_luma(rgb, esim=True):
    """[..., 3] -> [..., 1]; BT.601 weights for esim, BT.709 otherwise.  (utils/event_utils.py:23-33 builds the
    weight tensor on every call; here it is built once per device -- on a GPU that construction is a synchronous
    host-to-device copy in the middle of the step.)"""
    key = (rgb.device, bool(esim))
    factors = _luma_factors.get(key)
    if factors is None:
        w = (0.299, 0.587, 0.114) if esim else (0.2126, 0.7152, 0.0722)
        factors = _luma_factors[key] = torch.tensor(w, dtype=torch.float32, device=rgb.device)
    return torch.sum(rgb * factors[None, :], axis=-1)[..., None]


def lin_log(color, linlog_thres=20):
    """Linear below `linlog_thres`, natural log above, continuous at the threshold."""
    lin_slope = np.log(linlog_thres) / linlog_thres
    return torch.where(color < linlog_thres, lin_slope * color, torch.log(color))


class EventOptions:
    """The subset of the reference's CLI namespace the event step reads (main_nerf.py:97-185)."""

    def __init__(self, **kw):
        self.out_dim_color = 3
        self.use_luma = True
        self.linlog = True
        self.log_thres = 1e-7   # nerf/utils.py:349
        self.C_thres = 0.2
        self.event_only = True
        self.weight_loss_rgb = 1.0
        self.render_kwargs = {}
        # --negative_event_sampling (main_nerf.py; nerf/utils.py:340-345,548): off in every shipped config
        self.negative_event_sampling = False
        self.w_no_ev = 1.0
        self.epoch = 1
        self.epoch_start_noEvLoss = 0
        self.__dict__.update(kw)


def event_loss(image1, image2, pols, opt):
    """(lin-)log intensity change between two renders vs polarity; returns (loss, delta_linlog)."""
    if opt.use_luma:
        l1 = rgb_to_luma(image1, esim=True)
        l2 = rgb_to_luma(image2, esim=True)
    else:
        l1, l2 = image1, image2
    if opt.linlog:
        p1 = lin_log(l1 * 255, linlog_thres=20)
        p2 = lin_log(l2 * 255, linlog_thres=20)
    else:
        thres = torch.as_tensor(opt.log_thres, dtype=l1.dtype, device=l1.device)
        p1 = torch.log(torch.maximum(l1 * 255, thres))
        # the reference evaluates the second term on the *first* luma when use_luma is set (nerf/utils.py:500)
        p2 = torch.log(torch.maximum((l1 if opt.use_luma else l2) * 255, thres))
    delta = p2 - p1
    gt_pol = pols[..., None]
    w = 1.0
    if opt.C_thres != -1:
        loss = w * torch.mean((delta - gt_pol * opt.C_thres) ** 2)
    else:
        EPS = 1e-9
        w *= 20
        if not opt.event_only:
            w *= 20
        dn = delta / (torch.linalg.norm(delta, dim=1, keepdim=True) + EPS)
        pn = gt_pol / (torch.linalg.norm(gt_pol, dim=1, keepdim=True) + EPS)
        loss = w * torch.mean((dn - pn) ** 2)
    return loss, delta


def no_event_loss(image1, image2, opt):
    """The no-event term (nerf/utils.py:548-565): where no event fired between two times the (lin-)log intensity may not
    have changed by more than the contrast threshold: w_no_ev * mean(relu(|L2 - L1| - C)), C = C_thres if > 0 else 0.25.
    The reference uses lin_log here whatever `linlog` says."""
    if opt.use_luma:
        p1 = lin_log(rgb_to_luma(image1, esim=True) * 255, linlog_thres=20)
        p2 = lin_log(rgb_to_luma(image2, esim=True) * 255, linlog_thres=20)
    else:
        p1 = lin_log(image1 * 255, linlog_thres=20)
        p2 = lin_log(image2 * 255, linlog_thres=20)
    cno = opt.C_thres if opt.C_thres > 0 else 0.25
    return opt.w_no_ev * torch.mean(torch.relu(torch.abs(p2 - p1) - cno))


def wants_no_event_term(opt):
    return bool(getattr(opt, "negative_event_sampling", False)) and opt.epoch > opt.epoch_start_noEvLoss


def event_loss_with_grads(image1, image2, pols, opt):
    """-> (loss, delta, d loss / d image1, d loss / d image2).  On the device, fp32, C_thres != -1: one launch
    (enerf_event_loss_fwd_bwd, csrc/event_pairs.hip) in place of the ~50 elementwise / reduction launches of
    event_loss + autograd; anything else goes through those."""
    fused = (image1.is_cuda and opt.C_thres != -1 and image1.dtype == image2.dtype == pols.dtype == torch.float32
             and image1.shape == image2.shape and image1.shape[-1] == 3 and pols.numel() * 3 == image1.numel())
    if fused:
        from . import _lib as L
        a, b, p = image1.detach().contiguous(), image2.detach().contiguous(), pols.contiguous()
        n = p.numel()
        g1, g2 = torch.empty_like(a), torch.empty_like(b)
        delta = torch.empty(*a.shape[:-1], 1 if opt.use_luma else 3, dtype=torch.float32, device=a.device)
        loss = torch.empty((), dtype=torch.float32, device=a.device)
        L.check(L.lib().enerf_event_loss_fwd_bwd(a.data_ptr(), b.data_ptr(), p.data_ptr(), n, int(bool(opt.use_luma)),
                                                 int(bool(opt.linlog)), float(opt.C_thres), float(opt.log_thres), 1.0,
                                                 g1.data_ptr(), g2.data_ptr(), delta.data_ptr(), loss.data_ptr(),
                                                 L.stream_handle()), "event_loss_fwd_bwd")
        return loss, delta, g1, g2
    a = image1.detach().requires_grad_(True)
    b = image2.detach().requires_grad_(True)
    with torch.enable_grad():
        loss, delta = event_loss(a, b, pols, opt)
        g1, g2 = torch.autograd.grad(loss, [a, b], allow_unused=True)
    g1 = torch.zeros_like(a) if g1 is None else g1
    g2 = torch.zeros_like(b) if g2 is None else g2
    return loss.detach(), delta.detach(), g1, g2


def train_step_events(model, data, opt, criterion=None, bg_color=None, bg_color_no_evs=None):
    """One event training step's forward: two renders (+ optional frame render, + optional no-event pair of renders)
    -> loss.  nerf/utils.py:482-573.  `bg_color` / `bg_color_no_evs` [B,1,C] replace the step's random background draws
    (tests)."""
    images = data["images"]
    B = images.shape[0]
    dev = data["rays_evs_o1"].device
    # the reference draws this on the host and copies it over (nerf/utils.py:497); a pageable host->device copy
    # drains the stream every step, so the same U[0,1) draw is made on the device
    bg = torch.rand((B, 1, opt.out_dim_color), device=dev) if bg_color is None else bg_color
    kw = dict(opt.render_kwargs)
    kw.setdefault("out_dim_color", opt.out_dim_color)
    out1 = model.render(data["rays_evs_o1"], data["rays_evs_d1"], staged=False, bg_color=bg, perturb=True, **kw)
    out2 = model.render(data["rays_evs_o2"], data["rays_evs_d2"], staged=False, bg_color=bg, perturb=True, **kw)
    loss, delta = event_loss(out1["image"], out2["image"], data["pols"], opt)
    if not opt.event_only:
        C = images.shape[-1]
        if C == 4:
            bg_color = torch.rand_like(images[..., :opt.out_dim_color])
            gt = images[..., :opt.out_dim_color] * images[..., opt.out_dim_color:] + \
                bg_color * (1 - images[..., opt.out_dim_color:])
        else:
            bg_color, gt = None, images
        out = model.render(data["rays_o"], data["rays_d"], staged=False, bg_color=bg_color, perturb=True, **kw)
        crit = criterion if criterion is not None else torch.nn.MSELoss(reduction="none")
        loss = loss + opt.weight_loss_rgb * crit(out["image"], gt).mean()
    if wants_no_event_term(opt):
        # two more renders at pixels / times without events, a fresh background draw (nerf/utils.py:548-565)
        bg_no = torch.rand((B, 1, opt.out_dim_color), device=dev) if bg_color_no_evs is None else bg_color_no_evs
        n1 = model.render(data["rays_no_evs_o1"], data["rays_no_evs_d1"], staged=False, bg_color=bg_no, perturb=True, **kw)
        n2 = model.render(data["rays_no_evs_o2"], data["rays_no_evs_d2"], staged=False, bg_color=bg_no, perturb=True, **kw)
        loss = loss + no_event_loss(n1["image"], n2["image"], opt)
    return loss, delta


def train_step_events_manual(model, data, opt, after_forward=None, bg_color=None, defer_table=False,
                             after_mlp_backward=None, bg_color_no_evs=None):
    """The event-only step with the two renders driven without autograd (fused_render.render_train_raw /
    backward_raw): only the loss itself -- a few elementwise ops on two [N,3] images -- goes through autograd, and its
    gradient is handed to the renders' closed backward.  Same values as train_step_events + loss.backward()
    (nerf/utils.py:482-546); leaves the gradients in p.grad and returns (loss, delta).  `after_forward()` is called
    once both forwards are queued, `after_mlp_backward()` once the second render's MLP backward kernels are."""
    from . import fused_network as fnet
    from . import fused_render as fr
    B = data["images"].shape[0]
    dev = data["rays_evs_o1"].device
    bg = torch.rand((B, 1, opt.out_dim_color), device=dev) if bg_color is None else bg_color
    kw = {k: v for k, v in opt.render_kwargs.items() if k in ("dt_gamma", "max_steps")}
    shape = data["rays_evs_o1"].shape[:-1]
    img1, ctx1 = fr.render_train_raw(model, data["rays_evs_o1"], data["rays_evs_d1"], bg, True, **kw)
    img2, ctx2 = fr.render_train_raw(model, data["rays_evs_o2"], data["rays_evs_d2"], bg, True, **kw)
    ctxs = [ctx1, ctx2]
    no_ev = wants_no_event_term(opt)
    if no_ev:
        # the no-event pair of renders (nerf/utils.py:548-551): same closed-form route, its own background draw
        bg_no = torch.rand((B, 1, opt.out_dim_color), device=dev) if bg_color_no_evs is None else bg_color_no_evs
        nshape = data["rays_no_evs_o1"].shape[:-1]
        img3, ctx3 = fr.render_train_raw(model, data["rays_no_evs_o1"], data["rays_no_evs_d1"], bg_no, True, **kw)
        img4, ctx4 = fr.render_train_raw(model, data["rays_no_evs_o2"], data["rays_no_evs_d2"], bg_no, True, **kw)
        ctxs += [ctx3, ctx4]
    if after_forward is not None:
        after_forward()                                 # e.g. the next step's two marches on a side stream
    loss, delta, g1, g2 = event_loss_with_grads(img1.view(*shape, 3), img2.view(*shape, 3), data["pols"], opt)
    grads = [g1, g2]
    if no_ev:
        a = img3.view(*nshape, 3).detach().requires_grad_(True)
        b = img4.view(*nshape, 3).detach().requires_grad_(True)
        with torch.enable_grad():
            loss_no = no_event_loss(a, b, opt)
            g3, g4 = torch.autograd.grad(loss_no, [a, b])
        loss = loss + loss_no.detach()
        grads += [g3, g4]
    params = fnet.network_params(model)
    emb = params[0]
    keep = emb.grad if defer_table else None            # (deferred flush: the dense buffer is kept, and kept clean)
    for p in params:
        p.grad = None
    if defer_table:
        emb.grad = keep if keep is not None else torch.zeros_like(emb)
    # defer_table: every render's table gradients stay record lists for FusedAdam.step_grid_table (one flush)
    total = sum(c["M"] for c in ctxs) if defer_table else 0
    dw = None
    for i, (c, g) in enumerate(zip(ctxs, grads)):
        last = i == len(ctxs) - 1
        g_emb, dwi = fr.backward_raw(c, g_image=g, raw=True, defer_table=total,
                                     after_mlp=after_mlp_backward if last else None)
        if g_emb is not None:                           # the first backward's buffer; later ones add straight into it
            if emb.grad is None:                        # ... unless they could not (then they return their own)
                emb.grad = g_emb
            else:
                emb.grad.add_(g_emb)
        dw = dwi if dw is None else dw.add_(dwi)
    kind = ctx1["sv"]["kind"]
    for p, g in zip(params[1:], fnet.unpack_weight_grads(dw, ctx1["sv"]["out_c"], kind)):
        p.grad = g.view_as(p)
    return loss.detach(), delta
