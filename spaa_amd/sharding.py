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
    """ONE all-gather of every rank's result block back into global sample order.
    local_tensors: tuple of [n_local, ...] tensors of one dtype (cam_infer_best, prj_adv_best).  Returns a tuple of
    [n_total, ...] tensors on every rank.  The tensors of a rank are packed side by side into one row-major
    [maxn, F_0 + F_1 + ...] send block (zero rows up to the largest shard) and gathered by a single
    `all_gather_into_tensor` into a preallocated [world * maxn, F] buffer: one collective instead of one per tensor and
    no list outputs (on RCCL every peer writes its block over its own xGMI link; 2 x 50 MB per rank at B = 64 per GPU).
    `always_collective`: enter the collective even with one rank (tests: the RCCL path on a one-GPU box)."""
    if dist is None or not dist.is_initialized() or (dist.get_world_size() == 1 and not always_collective):
        return tuple(local_tensors)
    world = dist.get_world_size()
    ranges = [shard_range(n_total, r, world) for r in range(world)]
    maxn = max(hi - lo for lo, hi in ranges)
    first = local_tensors[0]
    feats = [int(torch.tensor(t.shape[1:]).prod()) if t.ndim > 1 else 1 for t in local_tensors]
    if any(t.dtype != first.dtype or t.shape[0] != first.shape[0] for t in local_tensors):
        raise ValueError('gather_results: the result tensors of a rank must share dtype and sample count')
    send = torch.zeros(maxn, sum(feats), dtype=first.dtype, device=first.device)
    col = 0
    for t, f in zip(local_tensors, feats):
        send[:t.shape[0], col:col + f] = t.reshape(t.shape[0], f)
        col += f
    recv = torch.empty(world * maxn, sum(feats), dtype=first.dtype, device=first.device)
    dist.all_gather_into_tensor(recv, send)
    even = maxn * world == n_total   # equal shards: the gathered rows are already in global order
    if not even:
        rows = torch.cat([torch.arange(r * maxn, r * maxn + hi - lo, device=first.device) for r, (lo, hi) in enumerate(ranges)])
    out, col = [], 0
    for t, f in zip(local_tensors, feats):
        blk = recv[:, col:col + f] if even else recv[rows, col:col + f]
        out.append(blk.reshape((n_total,) + tuple(t.shape[1:])))
        col += f
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
