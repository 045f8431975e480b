"""Checkpoints in the reference's format (row f4): the dict `Trainer.save_checkpoint` writes and `Trainer.load_checkpoint`
reads (nerf/utils.py:1295-1351, 1353-1415):

    {'epoch', 'global_step', 'stats', ['mean_count', 'mean_density' when cuda_ray],
     ['optimizer', 'lr_scheduler', 'scaler' when full], 'model': state_dict}

so that a run trained by the reference resumes on this path and the other way round.  The model's module / parameter /
buffer names are the reference's (tests/golden/ref_state_dict_schema.json); what needs care is the optimizer:
`FusedAdam` keeps `step` as a Python int and only the hyper-parameters it uses, `torch.optim.Adam` keeps `step` as a
float32 tensor and a dozen per-group flags that its `step()` reads -- the file always holds the torch form.
"""
import torch

# per-group keys torch.optim.Adam's step() reads (torch 2.x); written with Adam's defaults when the optimizer here does
# not carry them, so that `torch.optim.Adam.load_state_dict` + `step()` work on a file written from FusedAdam
_TORCH_ADAM_GROUP_DEFAULTS = dict(weight_decay=0, amsgrad=False, maximize=False, foreach=None, capturable=False,
                                  differentiable=False, fused=None, decoupled_weight_decay=False)


def _optimizer_state_for_file(opt, skipped=0):
    """`skipped`: optimizer steps the fp16 closed form's loss scaling did not apply (TrainHarness.amp_skipped_steps): the
    fused optimizer's host-side `step` counts them, torch's -- which the file holds -- does not."""
    sd = opt.state_dict()
    state = {}
    for k, st in sd["state"].items():
        st = dict(st)
        if "step" in st and not torch.is_tensor(st["step"]):
            st["step"] = torch.tensor(float(max(int(st["step"]) - int(skipped), 0)), dtype=torch.float32)
        state[k] = st
    groups = []
    for g in sd["param_groups"]:
        g = dict(g)
        for k, v in _TORCH_ADAM_GROUP_DEFAULTS.items():
            g.setdefault(k, v)
        groups.append(g)
    return {"state": state, "param_groups": groups}


def _load_optimizer_state(opt, sd):
    from .optim import FusedAdam
    opt.load_state_dict(sd)
    if isinstance(opt, FusedAdam):
        for st in opt.state.values():
            if torch.is_tensor(st.get("step")):
                st["step"] = int(round(float(st["step"])))
        opt._group_of = None


def checkpoint_dict(harness, full=False):
    """The reference's checkpoint dict for `harness` (a TrainHarness, or anything with .model, .opt, .global_step)."""
    m = harness.model
    state = {"epoch": int(getattr(harness, "epoch", 1)), "global_step": int(harness.global_step),
             "stats": getattr(harness, "stats", None) or {"loss": [], "valid_loss": [], "results": [],
                                                          "checkpoints": [], "best_result": None}}
    if m.cuda_ray:
        state["mean_count"] = int(m.mean_count)
        state["mean_density"] = float(m.mean_density)
    if full:
        gather = getattr(harness, "gather_sharded_optimizer_state", None)
        if gather is not None and getattr(harness, "comm_mode", None) == "sharded":
            gather()                      # the sharded data-parallel tail keeps 1/N of the table's moments per rank
        skipped = getattr(harness, "amp_skipped_steps", lambda: 0)()
        state["optimizer"] = _optimizer_state_for_file(harness.opt, skipped)
        sched = getattr(harness, "lr_scheduler", None)
        if sched is not None:
            state["lr_scheduler"] = sched.state_dict()
        scaler = getattr(harness, "scaler", None)
        state["scaler"] = scaler.state_dict() if scaler is not None else {}
    state["model"] = m.state_dict()
    return state


def save_checkpoint(harness, path, full=False):
    torch.save(checkpoint_dict(harness, full=full), path)
    return path


def _warn(what, err):
    import warnings
    warnings.warn(f"[checkpoint] failed to load {what}: {err!r} (continuing without it, as the reference's "
                  f"Trainer.load_checkpoint does, nerf/utils.py:1396-1415)")


def _numpy_scalar_globals():
    """What pickling a numpy scalar (np.float64(3.1)) refers to: the scalar constructor, the dtype class and its
    instances' types.  Names moved between numpy 1.x and 2.x; whatever exists is listed."""
    import numpy as np
    out = [np.dtype, np.float64, np.float32, np.int64, np.int32, np.bool_]
    for mod in ("numpy._core.multiarray", "numpy.core.multiarray"):
        try:
            out.append(getattr(__import__(mod, fromlist=["scalar"]), "scalar"))
        except Exception:                       # noqa: BLE001
            pass
    for name in ("Float64DType", "Float32DType", "Int64DType", "Int32DType", "BoolDType"):
        t = getattr(getattr(np, "dtypes", None), name, None)
        if t is not None:
            out.append(t)
    return out


def load_checkpoint(harness, checkpoint, model_only=False, map_location=None, trusted=False):
    """`checkpoint`: a path / file object, or an already loaded dict.  Mirrors Trainer.load_checkpoint: a bare
    state_dict is accepted; the model loads non-strictly and the (missing, unexpected) key lists are returned; the
    sample budget and mean density follow when the model marches on the occupancy grid; optimizer / scheduler / scaler
    are restored when present and wanted, and -- as in the reference (nerf/utils.py:1396-1415) -- one of those failing
    to load is a warning, not an error.
    Files are read with `weights_only=True`, with numpy's scalar reconstruction allow-listed: the reference-format dict
    is tensors, numbers, strings, lists and dicts, plus -- when the run kept a metric instead of the loss
    (`use_loss_as_metric=False`, nerf/utils.py:275-281,1278-1280) -- `numpy.float64` values of `PSNRMeter.measure()` in
    `stats['results']` / `stats['best_result']`.  A file that holds anything else fails with a hint; `trusted=True`
    allows the full unpickler for it."""
    m = harness.model
    if not isinstance(checkpoint, dict):
        dev = map_location or next(m.parameters()).device
        if trusted:
            checkpoint = torch.load(checkpoint, map_location=dev, weights_only=False)
        else:
            import pickle
            try:
                with torch.serialization.safe_globals(_numpy_scalar_globals()):
                    checkpoint = torch.load(checkpoint, map_location=dev, weights_only=True)
            except pickle.UnpicklingError as e:
                raise pickle.UnpicklingError(
                    f"{e}\n[enerf_amd.checkpoint] the file holds objects outside the weights-only allow-list (tensors, "
                    f"numbers, strings, containers, numpy scalars); if you trust its origin, load it with "
                    f"load_checkpoint(..., trusted=True)") from e
    if "model" not in checkpoint:
        m.load_state_dict(checkpoint)
        return [], []
    missing, unexpected = m.load_state_dict(checkpoint["model"], strict=False)
    if m.cuda_ray:
        # an update whose read-back is still in flight belongs to the run before the load: drop it, or the next read
        # of mean_density would resolve it over the loaded value (or raise from inside the property)
        if getattr(m, "_pending_density_stats", None) is not None:
            m._pending_density_stats = None
        if "mean_count" in checkpoint:
            m.mean_count = checkpoint["mean_count"]
        if "mean_density" in checkpoint:
            m.mean_density = checkpoint["mean_density"]
        # what was marched ahead, and the library's box of occupied cells, belong to the old bitfield
        m._premarched = None
        m.__dict__.pop("_native_ctx", None)       # (cached pointers of the one-call steps: parameters / moments may move)
        m.__dict__.pop("_native_events_ctx", None)
        from . import raymarching
        epoch = getattr(raymarching, "BITFIELD_EPOCH", None)
        if epoch is not None:
            epoch[0] += 1
    if model_only:
        return list(missing), list(unexpected)
    if "stats" in checkpoint:
        harness.stats = checkpoint["stats"]
    if "epoch" in checkpoint:
        harness.epoch = checkpoint["epoch"]
    if "global_step" in checkpoint:
        harness.global_step = checkpoint["global_step"]
    if getattr(harness, "opt", None) is not None and "optimizer" in checkpoint:
        try:
            _load_optimizer_state(harness.opt, checkpoint["optimizer"])
        except Exception as e:                  # noqa: BLE001
            _warn("optimizer", e)
        m.__dict__.pop("_native_ctx", None)
        m.__dict__.pop("_native_events_ctx", None)
        if hasattr(harness, "_cleared_grad"):
            harness._cleared_grad = None
        if getattr(harness, "_amp_words", None) is not None:
            harness._amp_words.zero_()            # (the loaded step counts are net of skipped steps)
    sched = getattr(harness, "lr_scheduler", None)
    if sched is not None and "lr_scheduler" in checkpoint:
        try:
            sched.load_state_dict(checkpoint["lr_scheduler"])
        except Exception as e:                  # noqa: BLE001
            _warn("scheduler", e)
    scaler = getattr(harness, "scaler", None)
    if scaler is not None and checkpoint.get("scaler"):
        try:
            scaler.load_state_dict(checkpoint["scaler"])
        except Exception as e:                  # noqa: BLE001
            _warn("scaler", e)
    return list(missing), list(unexpected)
