"""Child process of tests/test_gpu_parity.py::test_rccl_path_single_rank: the `nccl` (= RCCL) branch of the multi-GPU path
on a one-GPU box — process-group init on the device, `spaa_sharded` through its collective, bench.py's `gather_final`."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch.distributed as dist
    import bench
    from spaa_amd import synthetic as syn
    from spaa_amd.models import PCNet, WarpingNet
    from spaa_amd.classifier import Classifier
    from spaa_amd.projector_based_attack import spaa
    from spaa_amd.sharding import spaa_sharded

    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    os.environ.setdefault('MASTER_PORT', str(bench.free_port()))
    torch.cuda.set_device(0)
    dev = 'cuda:0'
    dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device(dev))
    assert dist.get_backend() == 'nccl'
    sz = (64, 64)
    sd = syn.pcnet_state_dict(0, cam_sz=sz, mask='rect')
    pc = PCNet(sd['mask'], WarpingNet(out_size=sz))
    pc.load_state_dict(sd)
    pc = pc.to(dev)
    clf = Classifier('resnet18', dev, state_dict=syn.resnet18_state_dict(2, logit_gain=20.0), input_sz=(56, 56))
    setup = dict(classifier_crop_sz=(60, 60), prj_brightness=0.5, prj_im_sz=sz)
    scenes = syn.scenes(1, 3, sz)
    targets = [204, 291, 129]
    cam, prj = spaa_sharded(pc, clf, None, targets, True, scenes, 5, 'camdE_caml2', dev, setup, dist=dist,
                            always_collective=True, iters=3)
    cam0, prj0 = spaa(pc, clf, None, targets, True, scenes, 5, 'camdE_caml2', dev, setup, iters=3)
    assert cam.shape == (3, 3, 64, 64) and torch.equal(cam, cam0) and torch.equal(prj, prj0), 'gathered results differ / out of order'
    ms, (cam_all, prj_all) = bench.gather_final(cam, prj, dist, 1, torch.cuda.synchronize)
    assert ms is not None and torch.equal(cam_all, cam) and torch.equal(prj_all, prj)
    dt, per_rank = bench.reduce_times(0.5, dist, 1, dev)
    assert dt == 0.5 and per_rank == [0.5]
    dist.barrier()
    dist.destroy_process_group()
    print('RCCL_SINGLE_RANK_OK')


if __name__ == '__main__':
    main()
