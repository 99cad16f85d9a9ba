"""bench.py — attack-iterations/sec of the fused SPAA loop (PCNet + ResNet-18 fwd/bwd + dE2000 loss + PGD step).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N ...

Workload = BASELINE.json configs[1]: batch 64 (8 synthetic 256x256 scenes x 8 targets) per GPU, ResNet-18, fp32,
stealth loss camdE_caml2, d_thr 5, targeted.  A "step" is one pass of the loop body
(/root/reference/src/python/projector_based_attack.py:264-328) over the batch, inputs resident in HBM.
N > 1: every rank attacks its own 64 samples (weak scaling, no data-path collective); the only exchange is the final
result gather, timed separately (`gather_ms`).  Prints ONE JSON line on rank 0.
"""
import argparse
import json
import re
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_F32_MFMA_TFLOPS = 157.3  # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
PEAK_BF16_MFMA_TFLOPS = 2516.0  # dense bf16 MFMA (16x the fp32 rate; "~2.5 PF dense" in MI355X_MICROARCH.md)


def build_attack(rank, batch, size, n_scenes, dev, classifier='resnet18'):
    from spaa_amd import synthetic as syn
    from spaa_amd.models import PCNet, WarpingNet
    from spaa_amd.classifier import Classifier
    from spaa_amd.projector_based_attack import AttackState

    sz = (size, size)
    sd = syn.pcnet_state_dict(0, cam_sz=sz, mask='ones')
    pc = PCNet(sd['mask'], WarpingNet(out_size=sz))
    pc.load_state_dict(sd)
    pc = pc.to(dev)
    csd = {'resnet18': syn.resnet18_state_dict, 'vgg16': syn.vgg16_state_dict,
           'inception_v3': syn.inception_v3_state_dict}[classifier](2, logit_gain=20.0)
    clf = Classifier(classifier, dev, state_dict=csd)
    per = batch // n_scenes
    scenes = syn.scenes(1 + 1000 * rank, n_scenes, sz)
    scene_b = scenes.repeat_interleave(per, dim=0)
    targets = (syn.IMAGENET10_TARGETS * 8)[:per] * n_scenes
    crop = (size - 16, size - 16)
    setup = dict(classifier_crop_sz=crop, prj_brightness=0.5, prj_im_sz=sz)
    st = AttackState(pc, clf, targets, scene_b, 'camdE_caml2', setup, dev)
    return st, sd, csd, setup, scenes, targets


def usable_cores():
    """Cores this process may actually use: affinity mask and cgroup CPU quota, not the host's core count."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except AttributeError:
        pass
    for path in ('/sys/fs/cgroup/cpu.max', '/sys/fs/cgroup/cpu/cpu.cfs_quota_us'):
        try:
            with open(path) as fh:
                parts = fh.read().split()
            if path.endswith('cpu.max'):
                if parts[0] != 'max':
                    n = min(n, max(1, int(int(parts[0]) / int(parts[1]))))
            else:
                q = int(parts[0])
                if q > 0:
                    with open('/sys/fs/cgroup/cpu/cpu.cfs_period_us') as fh2:
                        n = min(n, max(1, q // int(fh2.read())))
        except (OSError, ValueError, IndexError):
            pass
    return max(1, min(n, 64))


def log(msg):
    print(f'[bench] {msg}', file=sys.stderr, flush=True)


def cpu_baseline(sd, csd, setup, scenes, budget_b=16, iters=8):
    """Oracle (CPU restatement of the reference composition: per-iteration grid rebuild, two backward passes) timed on
    the host cores on a bounded sample: `budget_b` samples x `iters` iterations of the same workload."""
    sys.path.insert(0, os.path.join(ROOT, 'oracle'))
    import spaa_oracle as so
    from spaa_amd import synthetic as syn
    cores = usable_cores()
    torch.set_num_threads(cores)
    clf = so.OracleClassifier('resnet18', csd)
    tg = (syn.IMAGENET10_TARGETS * 2)[:budget_b]
    so.spaa(sd, clf, tg, True, scenes[:1], 5, 'camdE_caml2', setup, iters=1, per_batch_grid=True)  # warm-up
    t0 = time.time()
    so.spaa(sd, clf, tg, True, scenes[:1], 5, 'camdE_caml2', setup, iters=iters, per_batch_grid=True)
    dt = time.time() - t0
    scene_it_s = budget_b * iters / dt
    return {'value': scene_it_s / 64.0, 'unit': 'attack-iterations/s (batch-64 equivalent)', 'cores': cores,
            'kind': 'port', 'scene_iterations_per_s': scene_it_s,
            'sample': f'{budget_b} samples x {iters} iterations of the same 256x256 ResNet-18 workload on the host '
                      f'CPU ({dt:.1f} s), scaled by 1/64 to a batch-64 iteration'}


def pmc_traffic(tile):
    """HBM bytes per launch of the dominant kernel from the committed PMC passes (tools/pmc_traffic.py), or None."""
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'profiles', 'r01_pmc_traffic.json')
    if not os.path.exists(path):
        return None
    m = re.match(r'(x6d(?:16)?(?:co)?(?:a3)?|x6v\d|x6)_(\d+)x(\d+)(?:g(\d))?', tile)
    if not m:
        return None
    fam, bm, bn, g = m.groups()
    if fam.startswith('x6d'):  # tapconv_x6d_kernel<waves, BN, MFMA shape, coalesced epilogue, pixel stages>
        want = (f'tapconv_x6d_kernel<{int(bm) // 32}, {bn}, {16 if "16" in fam else 32}, '
                f'{"true" if "co" in fam else "false"}, {3 if fam.endswith("a3") else 2}>')
    else:
        want = f'tapconv_{fam}_kernel<{bm}, {bn}' + (f', {g}>' if g else '>')
    with open(path) as fh:
        for k, e in json.load(fh)['kernels'].items():
            if k.startswith(want):
                return round(e['hbm_bytes_per_launch'])
    return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--batch', type=int, default=64)
    ap.add_argument('--size', type=int, default=256)
    ap.add_argument('--classifier', default='resnet18', choices=['resnet18', 'vgg16', 'inception_v3'],
                    help='BASELINE.json configs[1] is resnet18 (the bench line); the others are extra data points')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--profile-out', default=None, help='write the per-layer tapconv timing table (JSON) here')
    args = ap.parse_args()

    # stdout carries exactly ONE line, the JSON result: anything libraries print there (RCCL's start-up banner does) is
    # sent to stderr instead
    sys.stdout.flush()
    json_out = os.fdopen(os.dup(1), 'w')
    os.dup2(2, 1)

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs a GPU: spaa_amd has no CPU fallback')
    torch.cuda.set_device(local_rank)
    dev = f'cuda:{local_rank}'
    dist = None
    if world > 1 or os.environ.get('SPAA_BENCH_FORCE_DIST'):  # (the switch rehearses the RCCL path with one rank)
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        dist.init_process_group('nccl', rank=rank, world_size=world, device_id=torch.device(dev))

    log('building attack state')
    st, sd, csd, setup, scenes, targets = build_attack(rank, args.batch, args.size, 8, dev, args.classifier)
    torch.cuda.synchronize()
    log('warmup')
    hp = dict(targeted=True, d_thr=5, adv_lr=2, col_lr=1, p_thresh=0.9)

    for _ in range(args.warmup):
        st.iteration(**hp)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    log('timed region')
    t0 = time.perf_counter()
    for _ in range(args.steps):
        st.iteration(**hp)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    log(f'{args.steps} steps in {dt:.3f}s')
    t = torch.tensor([dt], device=dev, dtype=torch.float64)
    if dist is not None:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt = float(t.item())

    # final result gather (the path's only exchange, once per attack): prj_adv_best + cam_infer_best of every rank
    gather_ms = None
    cam, prj = st.results()
    if dist is not None:
        torch.cuda.synchronize()
        dist.barrier()
        g0 = time.perf_counter()
        outs = [torch.empty_like(prj) for _ in range(world)]
        outs2 = [torch.empty_like(cam) for _ in range(world)]
        dist.all_gather(outs, prj)
        dist.all_gather(outs2, cam)
        torch.cuda.synchronize()
        gather_ms = (time.perf_counter() - g0) * 1e3

    # instrumented pass: HIP events around every tapconv launch (same stream) -> per-kernel roofline
    from spaa_amd import convplan
    roof, table = None, {}
    if rank == 0:
        convplan.PROFILE = []
        n_prof = 3
        for _ in range(n_prof):
            st.iteration(**hp)
        torch.cuda.synchronize()
        per_tile, per_layer = {}, {}
        for name, key, flops, e0, e1, tile_id, nbytes in convplan.PROFILE:
            ms = e0.elapsed_time(e1)
            # roofline groups = kernel instantiations (as rocprofv3 reports them): split-K launches of a tile belong to it
            # (persistent launches 48..54 run the same instantiations as 34, 36, 42, 44, 35, 37, 25)
            kern = {48: 34, 49: 36, 50: 42, 51: 44, 52: 35, 53: 37, 54: 25}.get(tile_id % 100, tile_id % 100)
            base = convplan.TILE_NAMES.get(kern, 'auto')
            tile = convplan.TILE_NAMES.get(tile_id % 100, 'auto') + (('_streamk' if tile_id // 100 == 9 else f'_splitk{tile_id // 100}') if tile_id >= 100 else '')
            a = per_tile.setdefault(base, [0.0, 0.0, 0, 0.0])
            a[0] += flops
            a[1] += ms
            a[2] += 1
            a[3] += nbytes
            b = per_layer.setdefault(name, [0.0, 0.0, 0, tile])
            b[0] += flops
            b[1] += ms
            b[2] += 1
        convplan.PROFILE = None
        tot_ms = sum(v[1] for v in per_tile.values())
        dom = max(per_tile, key=lambda k: per_tile[k][1])
        f, ms, n, nb = per_tile[dom]
        ach = f / (ms * 1e-3) / 1e12
        if dom.startswith('x6'):
            # fp32 emulated with six bf16 MFMAs per product group: the matrix-core ceiling for algorithmic fp32 FLOPs
            # is the dense bf16 peak / 6
            peak, kname = PEAK_BF16_MFMA_TFLOPS / 6.0, f'tapconv_{dom} (bf16x6-split MFMA implicit-GEMM, fp32-exact operands)'
        else:
            peak, kname = PEAK_F32_MFMA_TFLOPS, f'tapconv_kernel<{dom}> (fp32 MFMA implicit-GEMM conv/deconv/dgrad)'
        roof = {'kernel': kname, 'bound': 'mfma',
                'achieved': round(ach, 2), 'peak': round(peak, 1), 'unit': 'TFLOP/s',
                'frac': round(ach / peak, 4), 'traffic': pmc_traffic(dom), 'traffic_unit': 'bytes/launch',
                'traffic_source': 'profiles/r01_pmc_traffic.json (rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE, separate '
                                  'passes over this same command; bench.py cannot collect PMC itself)',
                'algorithmic_bytes_per_launch': round(nb / n),
                'peak_note': 'algorithmic fp32 FLOP/s; dense bf16 MFMA peak 2516 TF / 6 partial products for the x6 kernels, '
                             '157.3 TF for the fp32-MFMA kernels',
                'avg_launch_us': round(ms * 1e3 / n, 2), 'launches_per_step': n // n_prof,
                'flop_per_launch': f / n, 'share_of_conv_time': round(ms / tot_ms, 3),
                'all_tapconv_tflops': round(sum(v[0] for v in per_tile.values()) / (tot_ms * 1e-3) / 1e12, 2),
                'conv_ms_per_step': round(tot_ms / n_prof, 3)}
        table = {k: {'tile': v[3], 'gflop_per_launch': v[0] / v[2] / 1e9, 'us_per_launch': v[1] * 1e3 / v[2],
                     'tflops': v[0] / (v[1] * 1e-3) / 1e12} for k, v in per_layer.items()}
        if args.profile_out:
            with open(args.profile_out, 'w') as fh:
                json.dump({'per_tile': {k: {'flop': v[0], 'ms': v[1], 'launches': v[2], 'algorithmic_bytes': v[3]}
                                        for k, v in per_tile.items()},
                           'per_layer': table}, fh, indent=1)

    if rank == 0:
        value = world * args.steps / dt
        out = {
            'metric': 'attack-iterations/sec (PCNet+classifier fwd/bwd), 256x256 batch=64',
            'value': round(value, 3), 'unit': 'attack-iterations/s', 'n_gpus': world, 'steps': args.steps,
            'warmup': args.warmup, 'ms_per_step': round(dt / args.steps * 1e3, 3), 'higher_is_better': True,
            'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
            'config': {'workload': f'{dict(resnet18="configs[1]", inception_v3="configs[2]", vgg16="configs[4] classifier, SPAA loop")[args.classifier]}: batch={args.batch} ({8} scenes x {args.batch // 8} targets) '
                                   f'{args.size}x{args.size}, {args.classifier}, camdE_caml2, per GPU',
                       'global_batch': args.batch * world, 'parallelism': f'dp{world} (independent shards)'},
            'scene_iterations_per_s': round(value * args.batch, 1),
            'roofline': roof,
        }
        if gather_ms is not None:
            out['gather_ms'] = round(gather_ms, 3)
        if world == 1 and not args.no_cpu_baseline and args.classifier == 'resnet18':
            log(f'cpu baseline on {usable_cores()} cores')
            out['cpu_baseline'] = cpu_baseline(sd, csd, setup, scenes)
        else:
            out['cpu_baseline'] = None
        json_out.write(json.dumps(out) + '\n')
        json_out.flush()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
