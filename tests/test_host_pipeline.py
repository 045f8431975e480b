

def test_event_loss_with_grads_host_route_matches_autograd():
    """events.event_loss_with_grads off the device (and for the normalised loss, C_thres == -1, everywhere) goes through
    event_loss + autograd; same loss, delta and image gradients as differentiating event_loss directly."""
    import torch
    from enerf_amd.events import EventOptions, event_loss, event_loss_with_grads
    g = torch.Generator().manual_seed(1)
    a = torch.rand(1, 257, 3, generator=g).clamp(1e-3, 1.0)
    b = (a + 0.05 * torch.randn(1, 257, 3, generator=g)).clamp(1e-3, 1.0)
    pols = torch.randint(-2, 3, (1, 257), generator=g).float()
    for kw in (dict(C_thres=0.2, use_luma=True, linlog=True), dict(C_thres=-1, use_luma=False, linlog=True),
               dict(C_thres=0.2, use_luma=True, linlog=False)):
        opt = EventOptions(event_only=True, **kw)
        loss, delta, g1, g2 = event_loss_with_grads(a, b, pols, opt)
        ar, br = a.clone().requires_grad_(True), b.clone().requires_grad_(True)
        ref_loss, ref_delta = event_loss(ar, br, pols, opt)
        r1, r2 = torch.autograd.grad(ref_loss, [ar, br], allow_unused=True)
        assert torch.equal(delta, ref_delta.detach()) and float(loss) == float(ref_loss)
        assert torch.equal(g1, torch.zeros_like(a) if r1 is None else r1)
        assert torch.equal(g2, torch.zeros_like(b) if r2 is None else r2)
