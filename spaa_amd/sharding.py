"""Multi-GPU: one process per GPU, samples block-sharded, zero communication inside the loop.

Each (scene, target) sample of SPAA evolves independently (no BatchNorm in PCNet, classifier in eval mode,
per-sample gradient normalisation and masks: /root/reference/src/python/projector_based_attack.py:290-328), so the
batch is partitioned across ranks and the only exchange is ONE gather of the results at the end
(`torch.distributed` backend "nccl" = RCCL over xGMI on MI355X; "gloo" in the CPU tests).  The reference itself
has no distributed path (nn.DataParallel with one device, classifier.py:38-39).
"""
import torch


def shard_range(n, rank, world):
    """Contiguous block partition of n samples: ranks < n % world get one extra."""
    q, r = divmod(n, world)
    lo = rank * q + min(rank, r)
    return lo, lo + q + (1 if rank < r else 0)


def gather_results(local_tensors, n_total, dist=None, always_collective=False):
    """All-gather per-rank result blocks (possibly unequal sizes) back into global sample order.
    local_tensors: tuple of [n_local, ...] tensors.  Returns tuple of [n_total, ...] tensors on every rank.
    `always_collective`: enter the collective even with one rank (tests: the RCCL path on a one-GPU box)."""
    if dist is None or not dist.is_initialized() or (dist.get_world_size() == 1 and not always_collective):
        return tuple(local_tensors)
    world, rank = dist.get_world_size(), dist.get_rank()
    out = []
    for t in local_tensors:
        t = t.contiguous()
        maxn = max(shard_range(n_total, r, world)[1] - shard_range(n_total, r, world)[0] for r in range(world))
        pad = torch.zeros((maxn,) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
        pad[:t.shape[0]] = t
        bufs = [torch.empty_like(pad) for _ in range(world)]
        dist.all_gather(bufs, pad)   # direct peer exchange; 2 x ~100 MB per rank at B=64/GPU
        parts = []
        for r in range(world):
            lo, hi = shard_range(n_total, r, world)
            parts.append(bufs[r][:hi - lo])
        out.append(torch.cat(parts, 0))
    return tuple(out)


def spaa_sharded(pcnet, classifier, imagenet_labels, target_idx, targeted, cam_scene, d_thr, stealth_loss, device,
                 setup_info, dist=None, attack=None, always_collective=False, **kw):
    """`spaa()` over this rank's block of the batch, then one gather. `cam_scene`: [1|B,3,H,W].
    A rank whose block is empty (fewer samples than ranks) runs no attack and contributes zero-length blocks, so every
    rank still enters the collective.  `attack` replaces `spaa` (the CPU tests pass a stand-in)."""
    if attack is None:
        from .projector_based_attack import spaa as attack
    n = len(target_idx)
    world = dist.get_world_size() if (dist is not None and dist.is_initialized()) else 1
    rank = dist.get_rank() if world > 1 else 0   # (one rank, or no process group: the whole batch)
    lo, hi = shard_range(n, rank, world)
    while cam_scene.ndim < 4:
        cam_scene = cam_scene[None]
    if cam_scene.shape[0] not in (1, n):
        raise ValueError('cam_scene must hold 1 or len(target_idx) scenes')   # (raised on every rank, before the gather)
    scene = cam_scene if cam_scene.shape[0] == 1 else cam_scene[lo:hi]
    if hi > lo:
        cam, prj = attack(pcnet, classifier, imagenet_labels, list(target_idx[lo:hi]), targeted, scene, d_thr,
                          stealth_loss, device, setup_info, **kw)
    else:
        dev = torch.device(device)
        cam = torch.zeros((0, 3) + tuple(cam_scene.shape[-2:]), device=dev)
        prj = torch.zeros((0, 3) + tuple(setup_info['prj_im_sz']), device=dev)
    return gather_results((cam, prj), n, dist, always_collective)
