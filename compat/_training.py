"""`CNF.training_losses` for the reference's training scripts (train_lfm.py:154-183, train_lfm_t2i.py:190-204), host side only.

This package implements the sampling hot path: forward kernels, no backward.  SURVEY.md 8(b) lets `training_losses` "delegate to
a PyTorch fallback", and under the overlay the reference tree is on the path anyway -- so the loss is evaluated through the
REFERENCE's own U-ViT (libs/uvit.py:182 / libs/uvit_t2i.py:192, loaded beside this package's module by `_overlay.load_shadowed`),
built once per network over the SAME `nn.Parameter` objects as the MI355X module: the optimizer the script made from
`nnet.parameters()`, its EMA copy and `state_dict()` all keep seeing one set of tensors, gradients land on them, and the next
`decode()` repacks the updated weights (the module watches its parameters' versions).  Without the reference tree there is
nothing to delegate to and the stub's error stands.

Scope: single process, ENFORCED (`unwrap`): under DDP / `accelerate launch --multi_gpu` the gradient all-reduce is armed by the
wrapper's forward, which this path does not go through -- every rank would silently train its own diverging copy -- so a wrapped
network or an initialised process group of more than one rank raises instead.

Weights edited through `.data` (the reference's EMA update, tools/utils_uvit.py:109: `p_dest.data.mul_().add_()`) change neither a
parameter's version counter nor its storage, the two things the packed bf16 blob watches: every `training_losses` call therefore
advances a process-wide counter, and the overlay's `decode / encode / decode_fixadp` repack a network whose blob is older than it
(`refresh`), so sampling from `nnet_ema` in the training process sees the current weights."""
import torch

_TRAIN_EPOCH = [0]        # advanced by every training_losses call of this process


def note_training_step():
    _TRAIN_EPOCH[0] += 1


def refresh(net):
    """Repack `net`'s weights if any training step ran since they were packed (in-place `.data` edits are invisible otherwise)."""
    net = net.module if hasattr(net, "module") and not hasattr(net, "embed_dim") else net
    if _TRAIN_EPOCH[0] and net.__dict__.get("_packed_epoch") != _TRAIN_EPOCH[0] and hasattr(net, "invalidate_packed"):
        net.invalidate_packed()
        net.__dict__["_packed_epoch"] = _TRAIN_EPOCH[0]


def _cfg_of(net):
    kw = dict(img_size=net.img_size, patch_size=net.patch_size, in_chans=net.in_chans, embed_dim=net.embed_dim, depth=net.depth,
              num_heads=net.num_heads, mlp_ratio=net.hidden / net.embed_dim, qkv_bias=False, mlp_time_embed=False,
              # every reference config trains with activation checkpointing (configs/*.py: use_checkpoint=True): without it the twin keeps
              # all block activations and U-ViT-L runs out of memory at the reference's batch sizes.  conv / skip: the only values built
              use_checkpoint=bool(getattr(net, "use_checkpoint", False)), conv=True, skip=True)
    if type(net).__module__.endswith("uvit_t2i"):
        kw.update(clip_dim=net.clip_dim, num_clip_token=net.num_clip_token)
    else:
        kw.update(num_classes=net.num_classes)
    return kw


def reference_twin(net, overlay):
    """The reference's UViT over `net`'s own parameters, or None when the reference's module cannot be found."""
    twin = net.__dict__.get("_reference_twin")
    if twin is not None:
        return twin
    name = "libs.uvit_t2i" if type(net).__module__.endswith("uvit_t2i") else "libs.uvit"
    try:
        __import__("libs")
        ref = overlay.load_shadowed(name, overlay.OVERLAY[name])
    except ImportError:
        ref = None
    if ref is None or not hasattr(ref, "UViT"):
        return None
    # the throw-away init draws CPU random numbers: the script's RNG stream stays where it was (devices=[]: no CUDA generator is
    # saved -- the default would touch, and create a context on, every visible GPU)
    with torch.random.fork_rng(devices=[]):
        twin = ref.UViT(**_cfg_of(net))
    own = dict(net.named_parameters())
    theirs = [n for n, _ in twin.named_parameters()]
    if set(theirs) != set(own):
        raise RuntimeError(f"state_dict keys differ from the reference's: {sorted(set(theirs) ^ set(own))[:6]} ...")
    for n in theirs:
        mod = twin
        *path, leaf = n.split(".")
        for p in path:
            mod = getattr(mod, p)
        mod._parameters[leaf] = own[n]        # the SAME tensor: gradients and optimizer steps are shared
    twin.train(net.training)
    net.__dict__["_reference_twin"] = twin    # (not a submodule: parameters are not listed twice)
    return twin


def flow_matching_loss(velocity, x1, sigma_min):
    """The conditional flow-matching objective of flow_matching.py:88-100 for one batch of data latents x1: a point on the straight path
    from noise to data, x_t = t x1 + (1 - (1 - sigma) t) eps, whose target velocity is d x_t / dt = x1 - (1 - sigma) eps; squared error of
    `velocity(t, x_t)` against it, averaged per sample.  The two random draws come in the reference's order (eps first, then t), so a
    seeded script sees the same numbers."""
    eps = torch.randn_like(x1)
    t = torch.rand(x1.shape[0], device=x1.device, dtype=x1.dtype)
    tb = t.view(-1, 1, 1, 1)
    keep = 1.0 - sigma_min
    x_t = tb * x1 + (1.0 - keep * tb) * eps
    target = x1 - keep * eps
    err = velocity(t, x_t) - target
    return (err * err).mean(dim=(1, 2, 3))


def unwrap(net):
    """The U-ViT module behind `net`; refuses what this path cannot train correctly (see the module docstring)."""
    import torch.distributed as dist
    wrapped = hasattr(net, "module") and not hasattr(net, "embed_dim")
    multi = dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1
    if wrapped or multi:
        raise NotImplementedError(
            "compat training_losses is single-process: " + ("the network is wrapped (DistributedDataParallel / accelerate), " if wrapped else "")
            + ("torch.distributed runs " + str(dist.get_world_size()) + " ranks, " if multi else "")
            + "and the loss would bypass the wrapper's forward, so no gradient all-reduce would be armed and every rank would train its own "
              "copy.  Train with the reference's own modules (drop the overlay for training) and load the state_dict for sampling.")
    return net
