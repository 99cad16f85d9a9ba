"""CPU, world_size 2, gloo: the N>1 path (block sharding + one final gather) reassembles results in sample order."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from spaa_amd.sharding import shard_range, gather_results


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
