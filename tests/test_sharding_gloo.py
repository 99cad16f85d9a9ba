"""CPU, world_size 2, gloo: the N>1 path (block sharding + one final gather) reassembles results in sample order."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from spaa_amd.sharding import shard_range, gather_results, spaa_sharded


def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def _worker(rank, world, port, n_total, ret):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    lo, hi = shard_range(n_total, rank, world)
    # stand-in for the per-rank attack result: sample i carries the value i (what matters here is the exchange)
    cam = torch.arange(lo, hi, dtype=torch.float32).view(-1, 1, 1, 1).expand(-1, 3, 4, 4).contiguous()
    prj = cam * 2
    g_cam, g_prj = gather_results((cam, prj), n_total, dist)
    ok = (g_cam.shape[0] == n_total and torch.equal(g_cam[:, 0, 0, 0], torch.arange(n_total, dtype=torch.float32))
          and torch.equal(g_prj, g_cam * 2))
    ret[rank] = bool(ok)
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gather_uneven_and_even():
    for n_total in (7, 16):
        port = _free_port()
        with mp.Manager() as m:
            ret = m.dict()
            mp.spawn(_worker, args=(2, port, n_total, ret), nprocs=2, join=True)
            assert dict(ret) == {0: True, 1: True}


def _stub_attack(pcnet, classifier, labels, target_idx, targeted, cam_scene, d_thr, stealth_loss, device, setup_info):
    """Stand-in for spaa(): encodes which (target, scene) pair every sample saw, so the test can check the slicing."""
    assert len(target_idx) >= 1, 'an empty shard must not reach the attack'
    b = len(target_idx)
    scene = cam_scene.expand(b, -1, -1, -1) if cam_scene.shape[0] == 1 else cam_scene
    assert scene.shape[0] == b
    cam = scene + torch.tensor(target_idx, dtype=torch.float32).view(-1, 1, 1, 1)
    prj = torch.tensor(target_idx, dtype=torch.float32).view(-1, 1, 1, 1).expand(-1, 3, *setup_info['prj_im_sz']).contiguous()
    return cam.contiguous(), prj


def _sharded_worker(rank, world, port, n_total, per_sample_scene, ret):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    targets = [100 + i for i in range(n_total)]
    scenes = torch.arange(n_total if per_sample_scene else 1, dtype=torch.float32).view(-1, 1, 1, 1).expand(-1, 3, 4, 6) * 1e-3
    setup = dict(prj_im_sz=(2, 3))
    cam, prj = spaa_sharded(None, None, None, targets, True, scenes.contiguous(), 5, 'caml2', 'cpu', setup, dist=dist,
                            attack=_stub_attack)
    want_scene = scenes if per_sample_scene else scenes.expand(n_total, -1, -1, -1)
    want_cam = want_scene + torch.tensor(targets, dtype=torch.float32).view(-1, 1, 1, 1)
    ok = (cam.shape == (n_total, 3, 4, 6) and prj.shape == (n_total, 3, 2, 3) and torch.equal(cam, want_cam)
          and torch.equal(prj[:, 0, 0, 0], torch.tensor(targets, dtype=torch.float32)))
    ret[rank] = bool(ok)
    dist.barrier()
    dist.destroy_process_group()


def test_spaa_sharded_slices_targets_and_scenes():
    """spaa_sharded's own slicing (targets, per-sample scenes or one shared scene) with a stand-in attack: uneven split,
    even split, and FEWER samples than ranks (the empty rank contributes a zero-length block instead of hanging)."""
    for n_total, per_sample in ((5, True), (4, False), (1, True), (1, False)):
        port = _free_port()
        with mp.Manager() as m:
            ret = m.dict()
            mp.spawn(_sharded_worker, args=(2, port, n_total, per_sample, ret), nprocs=2, join=True)
            assert dict(ret) == {0: True, 1: True}, (n_total, per_sample, dict(ret))


def _b512_worker(rank, world, port, ret):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    torch.set_num_threads(1)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    n_total = 512
    targets = [1000 + i for i in range(n_total)]
    scenes = (torch.arange(n_total, dtype=torch.float32).view(-1, 1, 1, 1) * 1e-3).expand(-1, 3, 4, 6).contiguous()
    seen = {}

    def attack(pcnet, classifier, labels, target_idx, targeted, cam_scene, *a):
        seen['n'], seen['first'] = len(target_idx), target_idx[0]
        return _stub_attack(pcnet, classifier, labels, target_idx, targeted, cam_scene, *a)

    cam, prj = spaa_sharded(None, None, None, targets, True, scenes, 5, 'caml2', 'cpu', dict(prj_im_sz=(2, 3)), dist=dist, attack=attack)
    want_cam = scenes + torch.tensor(targets, dtype=torch.float32).view(-1, 1, 1, 1)
    ok = (seen['n'] == 64 and seen['first'] == 1000 + 64 * rank and cam.shape == (512, 3, 4, 6) and torch.equal(cam, want_cam)
          and torch.equal(prj[:, 0, 0, 0], torch.tensor(targets, dtype=torch.float32)))
    ret[rank] = bool(ok)
    dist.barrier()
    dist.destroy_process_group()


def test_batch_512_over_eight_ranks_keeps_global_order():
    """BASELINE.json configs[3] / configs[4] by shape: B = 512 over 8 ranks = 64 samples per rank (rank r takes samples
    64 r .. 64 r + 63 and its scenes), one gather, results in global sample order on every rank (gloo, stand-in attack)."""
    port = _free_port()
    with mp.Manager() as m:
        ret = m.dict()
        mp.spawn(_b512_worker, args=(8, port, ret), nprocs=8, join=True)
        assert dict(ret) == {r: True for r in range(8)}
