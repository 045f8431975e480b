

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


def test_architecture_caches_do_not_outlive_their_model():
    """The fused routes cache "which architecture is this model" -- on the model.  Keyed by id(model) in a module-level
    dict (as it was), a network.py model built right after a network_ff one was deleted could inherit the dead model's
    entry when the allocator handed out the same addresses: bench.py's legs hit it once in a few runs."""
    import gc
    from enerf_amd import fused_network
    from enerf_amd.network import NeRFNetwork as Linear
    from enerf_amd.network_ff import NeRFNetwork as FF
    for _ in range(8):
        a = FF(encoding="hashgrid", encoding_dir="sphere_harmonics", bound=1, cuda_ray=False)
        ka = fused_network.kind_of(a)
        ia = id(a)
        del a
        gc.collect()
        b = Linear(encoding="hashgrid", encoding_dir="sphere_harmonics", bound=1, cuda_ray=False, out_dim_color=3)
        assert ka == "ff" and fused_network.kind_of(b) == "linear", (ka, fused_network.kind_of(b), ia == id(b))
        assert "_fused_kind" in b.__dict__                            # the verdict lives on THIS model
        assert not hasattr(fused_network, "_arch_ok")
        del b
        gc.collect()
