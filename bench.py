"""bench.py — attack-iterations/sec of the fused SPAA loop (PCNet + ResNet-18 fwd/bwd + dE2000 loss + PGD step).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N ...

Workload = BASELINE.json configs[1]: batch 64 (8 synthetic 256x256 scenes x 8 targets) per GPU, ResNet-18, fp32,
stealth loss camdE_caml2, d_thr 5, targeted.  A "step" is one pass of the loop body
(/root/reference/src/python/projector_based_attack.py:264-328) over the batch, inputs resident in HBM.
N > 1: every rank attacks its own 64 samples (weak scaling, no data-path collective); the only exchange is the final
result gather, timed separately (`gather_ms`).  `--gpus N` without a torchrun environment starts the N ranks itself
(before anything touches the GPU).  Prints ONE JSON line on rank 0.
"""
import argparse
import json
import re
import os
import socket
import subprocess
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_F32_MFMA_TFLOPS = 157.3  # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
PEAK_BF16_MFMA_TFLOPS = 2516.0  # dense bf16 MFMA (16x the fp32 rate; "~2.5 PF dense" in MI355X_MICROARCH.md)
WINO_ALGORITHMIC_GAIN = 36.0 / 16.0  # Winograd F(2x2,3x3): 16 products per 2x2 output tile instead of 36
PEAK_HBM_TBS = 8.0  # HBM3E spec (MI355X_MICROARCH.md; 6.3 TB/s is what a float4 copy achieves)

# algorithmic bytes per scene-iteration at 256x256 (3-channel fp32 images, per-batch constants
# excluded), scaled by the pixel count for other sizes
MB_PER_SCENE_256 = {'warp_fwd': 1.57 + 2.36,    # grid_sample (read x, write x_w) + the rough input cat([s, x_w*s]) it also writes
                    'warp_bwd_gather': 1.57,   # read g, write g_x
                    'stealth_loss': 2.36,      # read cam_infer, scene; write gradient
                    'step_and_track': 3.9}     # read g, read/write x, conditional copies of x and cam_infer
# with the two entry layers fused (csrc/conv1pair.hip) the warp kernel writes x_w only, and the fused launch reads x_w and s
# (2 x 0.79 MB) and writes res1_s and x1 (2 x 2.10 MB: 128 x 128 x 32 fp32)
MB_WARP_FWD_NO_CAT, MB_CONV1_PAIR = 1.57, 1.57 + 4.19


def build_attack(rank, batch, size, n_scenes, dev, classifier='resnet18', storage='f32', attack='spaa'):
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
    if attack == 'perc_al':
        # BASELINE.json configs[4]: PerC_AL.adversary_projector (400 iterations; alpha schedules of a 400-iteration run)
        from spaa_amd.perc_al import PerC_AL, PerCALState
        att = PerC_AL(device=torch.device(dev), max_iterations=400, alpha_l_init=1, alpha_c_init=0.5, confidence=0, storage=storage)
        with torch.cuda.device(dev):
            st = PerCALState(att, clf, scene_b, torch.tensor(targets), 5.0, True, crop)
        it = [0]

        def step():
            st.iteration(it[0] % 400)
            it[0] += 1
        st.step, st.results = step, (lambda: (st.result(), st.result()))
        return st, sd, csd, setup, scenes, targets
    st = AttackState(pc, clf, targets, scene_b, 'camdE_caml2', setup, dev, storage=storage)
    hp = dict(targeted=True, d_thr=5, adv_lr=2, col_lr=1, p_thresh=0.9)
    st.step = lambda: st.iteration(**hp)
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
    the host cores on a bounded sample: `budget_b` samples x `iters` iterations of the bench workload, plus the two call
    shapes of BASELINE.json configs[0] (one scene: untargeted B=1 for 50 iterations; targeted K=10, 5 of its 50
    iterations)."""
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
    # configs[0]: the reference's own CPU-runnable case (projector_based_attack.py:107 and :120)
    t0 = time.time()
    so.spaa(sd, clf, [1], False, scenes[:1], 5, 'camdE_caml2', setup, iters=50, per_batch_grid=True)
    dt1 = time.time() - t0
    t0 = time.time()
    so.spaa(sd, clf, syn.IMAGENET10_TARGETS, True, scenes[:1], 5, 'camdE_caml2', setup, iters=5, per_batch_grid=True)
    dt10 = time.time() - t0
    return {'value': scene_it_s / 64.0, 'unit': 'attack-iterations/s (batch-64 equivalent)', 'cores': cores,
            'kind': 'port', 'scene_iterations_per_s': scene_it_s,
            'sample': f'{budget_b} samples x {iters} iterations of the same 256x256 ResNet-18 workload on the host '
                      f'CPU ({dt:.1f} s), scaled by 1/64 to a batch-64 iteration',
            'configs0': {'untargeted_B1_50it': {'seconds': round(dt1, 2), 'iterations_per_s': round(50 / dt1, 3)},
                         'targeted_K10': {'seconds': round(dt10, 2), 'iterations_timed': 5,
                                          'iterations_per_s': round(5 / dt10, 3),
                                          'note': '5 of the 50 iterations timed (every iteration costs the same)'}}}


def pmc_traffic(tile):
    """HBM bytes per launch of the dominant kernel from the committed PMC passes (tools/pmc_traffic.py), or None."""
    prof = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'profiles')
    # (the latest round's file: rNN_pmc_traffic.json)
    cands = sorted((f for f in (os.listdir(prof) if os.path.isdir(prof) else []) if re.fullmatch(r'r\d+_pmc_traffic\.json', f)), reverse=True)
    path = os.path.join(prof, cands[0]) if cands else None
    if path is None:
        return None, None
    m = re.match(r'(x6d(?:16)?(?:co)?(?:a3)?|x6v\d|x6)_(\d+)x(\d+)(?:g(\d))?', tile)
    if tile.startswith('wino'):
        # template arguments: N tile, kernel variant, 0, canvas / K-range form, two input tensors, waves per workgroup
        plain = tile.replace('_canvas', '').replace('_2src', '')
        bn, var, nw = (64, 2, 4) if '_8x32x64' in plain else (64, 2, 8) if plain.endswith('x64') else (128, 3, 8)
        want = f"wino_x6_kernel<{bn}, {var}, 0, {'true' if '_canvas' in tile else 'false'}, {'true' if '_2src' in tile else 'false'}, {nw}>"
    elif not m:
        return None, None
    else:
        fam, bm, bn, g = m.groups()
        if fam.startswith('x6d'):  # tapconv_x6d_kernel<waves, BN, MFMA shape, coalesced epilogue, pixel stages>
            want = (f'tapconv_x6d_kernel<{int(bm) // 32}, {bn}, {16 if "16" in fam else 32}, '
                    f'{"true" if "co" in fam else "false"}, {3 if fam.endswith("a3") else 2}>')
        else:
            want = f'tapconv_{fam}_kernel<{bm}, {bn}' + (f', {g}>' if g else '>')
    with open(path) as fh:
        for k, e in json.load(fh)['kernels'].items():
            if k.startswith(want):
                return round(e['hbm_bytes_per_launch']), os.path.relpath(path, ROOT)
    return None, None


# ---------------------------------------------------------------------------------------------------------------------
# multi-rank glue (launcher, timed region, result gather): independent of the GPU so that tests/ can rehearse it on gloo
def free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def launch_ranks(n, argv):
    """`python bench.py --gpus N` outside torchrun: start the N ranks as CHILD processes (torch.distributed.run) and
    return its exit code.  Called before this process has made any GPU call; nothing is exec'd."""
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={n}', '--master-addr',
           '127.0.0.1', '--master-port', str(free_port()), os.path.abspath(__file__)] + list(argv)
    log('launching: ' + ' '.join(cmd))
    return subprocess.call(cmd)


def timed_steps(step, steps, warmup, dist, sync):
    """W untimed steps, then exactly K steps bracketed by barrier + device sync on both sides; seconds of THIS rank.
    The host's cyclic garbage collector is off between the two brackets (as `timeit` does): the loop allocates nothing that needs it, and a
    full collection over this process's 175 000 tracked objects takes 55 ms (tools/lab/gc_probe.py) -- inside a 20-step window of 80-160 ms
    that is the whole measurement (one run of this round read 6.16 ms per step where the same run's kernels summed to 3.88)."""
    import gc
    gc.collect()               # (BEFORE the warm-up steps: 55 ms of idle GPU right in front of the timed window would cost its first steps their clocks)
    gc_was_on = gc.isenabled()
    gc.disable()
    try:
        for _ in range(warmup):
            step()
        sync()
        if dist is not None:
            dist.barrier()
        sync()
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        sync()
        if dist is not None:
            dist.barrier()
        sync()
        return time.perf_counter() - t0
    finally:
        if gc_was_on:
            gc.enable()


def reduce_times(dt, dist, world, dev):
    """(max over ranks, per-rank list) of the timed region."""
    if dist is None:
        return dt, [dt]
    t = torch.tensor([dt], device=dev, dtype=torch.float64)
    every = [torch.zeros_like(t) for _ in range(world)]
    dist.all_gather(every, t)
    per_rank = [float(e.item()) for e in every]
    return max(per_rank), per_rank


def gather_final(cam, prj, dist, world, sync):
    """The path's only exchange: every rank's (cam_infer_best, prj_adv_best), packed side by side into one send block and
    gathered by ONE `all_gather_into_tensor` (spaa_amd/sharding.py: gather_results does the same).  Send and receive
    buffers are allocated BEFORE the timed gather; the packing copy is inside it."""
    if dist is None:
        return None, (cam, prj)
    n, fc, fp = cam.shape[0], cam[0].numel(), prj[0].numel()
    send = torch.empty(n, fc + fp, dtype=cam.dtype, device=cam.device)
    recv = torch.empty(world * n, fc + fp, dtype=cam.dtype, device=cam.device)
    sync()
    dist.barrier()
    g0 = time.perf_counter()
    send[:, :fc] = cam.reshape(n, fc)
    send[:, fc:] = prj.reshape(n, fp)
    dist.all_gather_into_tensor(recv, send)
    sync()
    ms = (time.perf_counter() - g0) * 1e3
    return ms, (recv[:, :fc].reshape((world * n,) + tuple(cam.shape[1:])), recv[:, fc:].reshape((world * n,) + tuple(prj.shape[1:])))


def rehearse_glue(args, world, rank, json_out):
    """CPU rehearsal of the N-rank glue (launcher -> rendezvous -> barrier-bracketed timed region -> MAX over ranks ->
    preallocated result gather -> one line on rank 0) on gloo with a stand-in step.  NOT a measurement: `value` is null."""
    import torch.distributed as dist
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    os.environ.setdefault('MASTER_PORT', str(free_port()))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    state = torch.zeros(4, 3, 8, 8)

    def step():
        state.add_(1.0)

    dt = timed_steps(step, args.steps, args.warmup, dist, lambda: None)
    dt_max, per_rank = reduce_times(dt, dist, world, 'cpu')
    gather_ms, (cam_all, prj_all) = gather_final(state + rank, state * 2 + rank, dist, world, lambda: None)
    ok = (cam_all.shape[0] == 4 * world and
          all(float(cam_all[4 * r, 0, 0, 0]) == args.steps + args.warmup + r for r in range(world)))
    if rank == 0:
        json_out.write(json.dumps({'metric': 'attack-iterations/sec (PCNet+classifier fwd/bwd), 256x256 batch=64',
                                   'value': None, 'rehearsal': 'multi-rank glue only (gloo, stand-in step): not a measurement',
                                   'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
                                   'per_rank_ms_per_step': [round(t / args.steps * 1e3, 4) for t in per_rank],
                                   'ms_per_step': round(dt_max / args.steps * 1e3, 4), 'gather_ms': round(gather_ms, 3),
                                   'gather_ok': bool(ok)}) + '\n')
        json_out.flush()
    dist.barrier()
    dist.destroy_process_group()


# ---------------------------------------------------------------------------------------------------------------------
def tile_names(tile_id):
    """(kernel instantiation as rocprofv3 reports it, tile label of the per-layer table) of a reported tile id (spaa_amd/convplan.py:
    tile + 100 x K ranges (9 = stream-K); Winograd: + 1000 canvas / K-range form, + 2000 two input tensors)."""
    from spaa_amd import convplan
    form, tid = tile_id // 1000, tile_id % 1000
    # (persistent launches 48..54 run the same instantiations as 34, 36, 42, 44, 35, 37, 25)
    kern = {48: 34, 49: 36, 50: 42, 51: 44, 52: 35, 53: 37, 54: 25}.get(tid % 100, tid % 100)
    suffix = {1: '_canvas', 2: '_2src'}.get(form, '')
    base = convplan.TILE_NAMES.get(kern, 'auto') + suffix
    tile = convplan.TILE_NAMES.get(tid % 100, 'auto') + suffix + (('_streamk' if tid // 100 == 9 else f'_splitk{tid // 100}') if tid >= 100 else '')
    return base, tile


def instrumented_pass(st, args, n_prof=3):
    """Per-kernel event timing (HIP events on the launch stream): every tapconv launch, then every other entry point."""
    from spaa_amd import convplan, _lib
    convplan.PROFILE = []
    for _ in range(n_prof):
        st.step()
    torch.cuda.synchronize()
    conv_events = convplan.PROFILE
    convplan.PROFILE = None
    per_tile, per_layer = {}, {}
    t_roof_x6 = t_roof_f32 = 0.0
    for name, key, flops, e0, e1, tile_id, nbytes in conv_events:
        ms = e0.elapsed_time(e1)
        # roofline groups = kernel instantiations (as rocprofv3 reports them): split-K launches of a tile belong to it
        base, tile = tile_names(tile_id)
        a = per_tile.setdefault(base, [0.0, 0.0, 0, 0.0])
        a[0] += flops
        a[1] += ms
        a[2] += 1
        a[3] += nbytes
        b = per_layer.setdefault(name, [0.0, 0.0, 0, tile, 0.0])
        b[0] += flops
        b[1] += ms
        b[2] += 1
        b[4] += nbytes
        t_hbm = nbytes / (PEAK_HBM_TBS * 1e12)
        # (a Winograd launch executes 16 of the 36 products of a 3 x 3 tap set: its ceiling for ALGORITHMIC FLOPs is 36/16 of the
        # direct bf16x6 kernels' -- each layer is priced against the ceiling of the kernel that runs it, the saving is credited once)
        peak_x6 = PEAK_BF16_MFMA_TFLOPS / 6.0 * (WINO_ALGORITHMIC_GAIN if base.startswith('wino') else 1.0)
        if args.dtype == 'f16s':   # (fp16 storage: every convolution launch is one fp16 MFMA per product -- the dense 16-bit peak;
            peak_x6 = PEAK_BF16_MFMA_TFLOPS   # rounds 3-6 priced these launches at the bf16x6 ceiling: step.frac > 1 on the f16s line)
        t_roof_x6 += max(t_hbm, flops / (peak_x6 * 1e12)) / n_prof
        t_roof_f32 += max(t_hbm, flops / (PEAK_F32_MFMA_TFLOPS * 1e12)) / n_prof
    # second pass: the non-convolution entry points
    _lib.PROFILE = []
    for _ in range(n_prof):
        st.step()
    torch.cuda.synchronize()
    other = {}
    for name, e0, e1 in _lib.PROFILE:
        if name == 'spaa_tapconv_f32':
            continue
        o = other.setdefault(name, [0.0, 0])
        o[0] += e0.elapsed_time(e1)
        o[1] += 1
    _lib.PROFILE = None
    return per_tile, per_layer, other, t_roof_x6, t_roof_f32


PCNET_LAYERS = ('conv1', 'conv2', 'conv3', 'conv4', 'conv5', 'conv1_s', 'conv2_s', 'conv3_s', 'conv4_s', 'conv6', 'skipConv2',
                'skipConv3', 'transConv1', 'transConv2')
# (round 6: spaa_warp_fwd_taps = grid_sample from the tap table; spaa_warp_bwd_tiled_sumsq = its adjoint WITH spaa_grad_sumsq in the
# epilogue -- counted in full although the ||g||^2 part is the attack step's, not PCNet's)
PCNET_ENTRY_POINTS = ('spaa_conv1_pair_fwd', 'spaa_conv1_pair_bwd_f16', 'spaa_fs2_h16', 'spaa_s2f_h16', 'spaa_s2f_x6', 'spaa_warp_fwd', 'spaa_warp_fwd_taps', 'spaa_warp_bwd_gather', 'spaa_warp_bwd_tiled', 'spaa_warp_bwd_tiled_sumsq',
                      'spaa_shading_tail_fwd', 'spaa_shading_tail_fwd_f16', 'spaa_shading_tail_fwd_g', 'spaa_shading_tail_fwd_f16_g',
                      'spaa_shading_head_bwd_select', 'spaa_shading_head_bwd_select_f16', 'spaa_shading_head_bwd_select_g', 'spaa_shading_head_bwd_select_f16_g',
                      'spaa_shading_head_bwd', 'spaa_shading_head_bwd_f16', 'spaa_stealth_loss_fwd_bwd')
MB_PCNET_DE_PER_SCENE_256 = 214.0   # PCNet + dE2000 forward / backward, fp32, per scene-iteration


def pcnet_de_hbm(per_layer, other, n_prof, batch, size, f16):
    """north_star's HBM framing: algorithmic bytes of PCNet + dE2000 forward / backward per step (214 MB per scene-
    iteration in fp32; the network's activations and gradients are half of that in fp16 storage, the 3-channel images stay
    fp32) over the time of exactly those kernels, as a fraction of 8 TB/s."""
    ms = 0.0
    for name, v in per_layer.items():
        base = name.split('+')[0]      # (fused launches: 'transConv1+skipConv2', 'conv2_dgrad+skipConv2_dgrad')
        base = base[:-6] if base.endswith('_dgrad') else base
        if base in PCNET_LAYERS:
            ms += v[1] / n_prof
    ms_head_select = 0.0
    for name in PCNET_ENTRY_POINTS:
        if name in other:
            ms += other[name][0] / n_prof
            if name.startswith('spaa_shading_head_bwd_select'):
                ms_head_select += other[name][0] / n_prof
    img_mb = 1.57 * 2 + 2.36          # warp forward / backward and the loss: fp32 images in either mode
    mb = img_mb + (MB_PCNET_DE_PER_SCENE_256 - img_mb) * (0.5 if f16 else 1.0)
    gb = mb * 1e6 * batch * (size * size) / 65536.0
    out = {'kernels_ms_per_step': round(ms, 3), 'algorithmic_bytes_per_step': round(gb),
           'achieved_tb_s': round(gb / (ms * 1e-3) / 1e12, 3), 'frac_of_8_tb_s': round(gb / (ms * 1e-3) / 1e12 / PEAK_HBM_TBS, 4)}
    if ms_head_select > 0.0:
        # Until round 5 the kernel list missed the SELECT form of the backward head (the entry point the loop has called since round 4):
        # rounds 4 and 5 reported this fraction WITHOUT that kernel's time (r05: 0.3735 / 0.3419).  Round 6 counts it; the old basis is
        # kept beside it so that the rounds can be compared
        ms5 = ms - ms_head_select
        out['head_select_ms'] = round(ms_head_select, 3)
        out['frac_of_8_tb_s_r05_basis'] = round(gb / (ms5 * 1e-3) / 1e12 / PEAK_HBM_TBS, 4)
        out['note'] = ('frac_of_8_tb_s counts every PCNet / dE2000 kernel of the step; frac_of_8_tb_s_r05_basis leaves out the select form of '
                       'the backward head, which the kernel list of rounds 4-5 missed (r05 lines: 0.3735 f16 storage, 0.3419 f32)')
    return out


def configs0_gpu(sd, csd, setup, scenes, dev):
    """BASELINE.json configs[0] on the GPU, next to the CPU leg's `configs0`: ONE scene, the reference's two calls
    (projector_based_attack.py:107 untargeted B = 1, :120 targeted K = 10), 50 iterations each through `spaa()` itself (state
    build, HIP-graph capture and replay, result conversion included)."""
    from spaa_amd import synthetic as syn
    from spaa_amd.models import PCNet, WarpingNet
    from spaa_amd.classifier import Classifier
    from spaa_amd.projector_based_attack import spaa, AttackState
    sz = tuple(setup['prj_im_sz'])
    pc = PCNet(sd['mask'], WarpingNet(out_size=sz))
    pc.load_state_dict(sd)
    pc = pc.to(dev)
    clf = Classifier('resnet18', dev, state_dict=csd)
    out = {}
    for name, tg, targeted in (('untargeted_B1_50it', [1], False), ('targeted_K10_50it', list(syn.IMAGENET10_TARGETS), True)):
        spaa(pc, clf, None, tg, targeted, scenes[:1], 5, 'camdE_caml2', dev, setup, iters=4)     # warm-up: engines, kernel attributes
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        cam, prj = spaa(pc, clf, None, tg, targeted, scenes[:1], 5, 'camdE_caml2', dev, setup, iters=50)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        # host time to ENQUEUE one eagerly launched iteration (what the graph replay removes)
        st = AttackState(pc, clf, tg, scenes[:1], 'camdE_caml2', setup, dev)
        st.iteration(targeted, 5, 2, 1, 0.9)
        torch.cuda.synchronize()
        h0 = time.perf_counter()
        for _ in range(10):
            st.iteration(targeted, 5, 2, 1, 0.9)
        h1 = time.perf_counter()
        torch.cuda.synchronize()
        h2 = time.perf_counter()
        from spaa_amd import projector_based_attack as _pba
        replayed = bool(_pba.LAST_RUN.get('graph'))
        out[name] = {'seconds': round(dt, 4), 'iterations_per_s': round(50 / dt, 2), 'graph_replay': replayed,
                     'eager_host_enqueue_ms_per_iteration': round((h1 - h0) / 10 * 1e3, 3),
                     'eager_ms_per_iteration': round((h2 - h0) / 10 * 1e3, 3)}
        del st
    graphs = [v['graph_replay'] for v in out.values()]
    out['note'] = ('spaa() end to end (50 iterations: 1 eager + 49 graph replays; the capture itself executes nothing); eager_* = the same loop body launched kernel by kernel'
                   if all(graphs) else 'spaa() end to end; GRAPH CAPTURE WAS REFUSED (graph_replay false): these runs launched every iteration kernel by kernel')
    return out


def time_mode(dev, args, classifier, storage, attack, steps=10, warmup=3):
    """One extra configuration of BASELINE.json timed like the headline (own state, own warm-up, own region); never `value`."""
    t_build = time.perf_counter()
    st, _sd, _csd, _setup, _scenes, _targets = build_attack(0, args.batch, args.size, 8, dev, classifier, storage, attack)
    torch.cuda.synchronize()
    dt = timed_steps(st.step, steps, warmup, None, torch.cuda.synchronize)
    out = {'attack_iterations_per_s': round(steps / dt, 3), 'ms_per_step': round(dt / steps * 1e3, 3), 'steps': steps, 'warmup': warmup,
           'batch': args.batch, 'size': args.size, 'classifier': classifier,
           'dtype': 'f32' if storage == 'f32' else 'f16 storage (activations/gradients), f32 accumulation, images and dE2000',
           'attack': 'spaa loop body' if attack == 'spaa' else 'PerC_AL.adversary_projector loop body'}
    if attack == 'spaa':
        per_tile, per_layer, other, _a, _b = instrumented_pass(st, args, 2)
        out['pcnet_dE_hbm'] = pcnet_de_hbm(per_layer, other, 2, args.batch, args.size, storage == 'f16')
        tot = sum(v[1] for v in per_tile.values())
        out['all_tapconv_tflops'] = round(sum(v[0] for v in per_tile.values()) / (tot * 1e-3) / 1e12, 2)
    out['seconds_incl_build'] = round(time.perf_counter() - t_build, 1)
    del st
    torch.cuda.empty_cache()
    return out


def time_sustained(dev, args, classifier, storage, attack, steps):
    """A configuration of BASELINE.json at ITS OWN iteration count (configs[1]: 200, configs[4]: 400) in one back-to-back run,
    with the rate of every block of 20 steps (HIP events recorded on the launch stream, no host sync inside the run): shows
    whether the 20-step headline rate holds once clocks and temperature have settled (MI355X_MICROARCH.md, DVFS give-back)."""
    st, _sd, _csd, _setup, _scenes, _targets = build_attack(0, args.batch, args.size, 8, dev, classifier, storage, attack)
    for _ in range(3):
        st.step()
    blocks = steps // 20
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(blocks + 1)]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(blocks * 20):
        if i % 20 == 0:
            evs[i // 20].record()
        st.step()
    evs[blocks].record()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    per20 = [round(evs[i].elapsed_time(evs[i + 1]) / 20.0, 3) for i in range(blocks)]
    del st
    torch.cuda.empty_cache()
    return {'steps': blocks * 20, 'attack_iterations_per_s': round(blocks * 20 / dt, 3), 'ms_per_step': round(dt / (blocks * 20) * 1e3, 3),
            'ms_per_step_first_20': per20[0], 'ms_per_step_last_20': per20[-1], 'ms_per_step_slowest_20': max(per20),
            'ms_per_step_per_20_steps': per20, 'batch': args.batch, 'classifier': classifier,
            'dtype': 'f32' if storage == 'f32' else 'f16 storage', 'attack': attack}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--batch', type=int, default=64)
    ap.add_argument('--size', type=int, default=256)
    ap.add_argument('--classifier', default='resnet18', choices=['resnet18', 'vgg16', 'inception_v3'],
                    help='BASELINE.json configs[1] is resnet18 (the bench line); the others are extra data points')
    ap.add_argument('--dtype', default='f32', choices=['f32', 'f16s'],
                    help='f32 (headline) or f16s = fp16 STORAGE of activations/gradients, fp32 accumulation / images / dE2000 '
                         '(BASELINE.json configs[4]; an extra data point, never the headline)')
    ap.add_argument('--attack', default='spaa', choices=['spaa', 'perc_al'],
                    help='spaa (the bench line) or perc_al = PerC_AL.adversary_projector loop body (configs[4], with --classifier vgg16)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-modes', action='store_true',
                    help='skip the extra configurations timed after the headline (f16 storage, Inception-v3, VGG-16 + PerC-AL)')
    ap.add_argument('--profile-out', default=None, help='write the per-layer tapconv timing table (JSON) here')
    ap.add_argument('--rehearse-glue', action='store_true',
                    help='CPU/gloo rehearsal of the multi-rank glue with a stand-in step (tests/); not a measurement')
    args = ap.parse_args()

    world_env = os.environ.get('WORLD_SIZE')
    if args.gpus > 1 and world_env is None:
        # not under torchrun: start the ranks ourselves, as children, BEFORE any GPU call in this process
        raise SystemExit(launch_ranks(args.gpus, sys.argv[1:]))
    world = int(world_env or '1')
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if world != args.gpus:
        raise SystemExit(f'bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: launch one rank per GPU '
                         f'(python -m torch.distributed.run --nproc-per-node {args.gpus} bench.py --gpus {args.gpus} ...)')

    # stdout carries exactly ONE line, the JSON result: anything libraries print there (RCCL's start-up banner does) is
    # sent to stderr instead
    sys.stdout.flush()
    json_out = os.fdopen(os.dup(1), 'w')
    os.dup2(2, 1)

    if args.rehearse_glue:
        return rehearse_glue(args, world, rank, json_out)

    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs a GPU: spaa_amd has no CPU fallback')
    torch.cuda.set_device(local_rank)
    dev = f'cuda:{local_rank}'
    dist = None
    if world > 1 or os.environ.get('SPAA_BENCH_FORCE_DIST'):  # (the switch rehearses the RCCL path with one rank)
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', str(free_port()))
        dist.init_process_group('nccl', rank=rank, world_size=world, device_id=torch.device(dev))

    log('building attack state')
    storage = 'f16' if args.dtype == 'f16s' else 'f32'
    st, sd, csd, setup, scenes, targets = build_attack(rank, args.batch, args.size, 8, dev, args.classifier, storage, args.attack)
    torch.cuda.synchronize()
    # which conv kernel dominates?  One instrumented step (outside the timed region) tells; its launches then carry HIP events
    # INSIDE the timed region (a dozen launches per step: nothing measurable on the step time), so that the roofline's launch
    # duration is that of the back-to-back loop the rocprofv3 summary of the same command sees
    from spaa_amd import convplan
    dom_ids, timed_events = None, []
    if rank == 0:
        st.step()
        convplan.PROFILE = []
        st.step()
        torch.cuda.synchronize()
        per = {}
        for _n, _k, _f, e0, e1, tid, _b in convplan.PROFILE:
            per[tid] = per.get(tid, 0.0) + e0.elapsed_time(e1)
        names = {}
        for tid, ms in per.items():
            names.setdefault(tile_names(tid)[0], []).append((tid, ms))
        dom_name = max(names, key=lambda k: sum(m for _, m in names[k]))
        dom_ids = {tid for tid, _ in names[dom_name]}
        convplan.PROFILE, convplan.PROFILE_ONLY = None, dom_ids
    # (the events ride on every FOURTH step of the timed region -- 35 of its 140 launches of the dominant kernel at K = 20 --: on every
    # step they cost 0.5 % of the step time, measured with SPAA_BENCH_NO_TIMED_EVENTS=1; the rocprofv3 average covers all launches)
    events_on = dom_ids is not None and os.environ.get('SPAA_BENCH_NO_TIMED_EVENTS') != '1'
    step_no = [0]

    def step_sampled():
        i = step_no[0] - args.warmup
        step_no[0] += 1
        convplan.PROFILE = timed_events if (events_on and i >= 0 and i % 4 == 0) else None
        st.step()
    log('warmup + timed region')
    dt_local = timed_steps(step_sampled, args.steps, args.warmup, dist, torch.cuda.synchronize)
    convplan.PROFILE, convplan.PROFILE_ONLY = None, None
    log(f'{args.steps} steps in {dt_local:.3f}s')
    dt, per_rank = reduce_times(dt_local, dist, world, dev)

    # final result gather (the path's only exchange, once per attack): prj_adv_best + cam_infer_best of every rank
    cam, prj = st.results()
    gather_ms, _ = gather_final(cam, prj, dist, world, torch.cuda.synchronize)

    # instrumented passes: HIP events around every launch (same stream) -> per-kernel rooflines
    from spaa_amd import convplan
    roof, table = None, {}
    if rank == 0:
        n_prof = 3
        per_tile, per_layer, other, t_roof_x6, t_roof_f32 = instrumented_pass(st, args, n_prof)
        tot_ms = sum(v[1] for v in per_tile.values())
        dom = max(per_tile, key=lambda k: per_tile[k][1])
        f, ms, n, nb = per_tile[dom]
        ms_instr, n_instr = ms, n
        if timed_events:   # the dominant kernel's launches of every fourth step of the timed region
            ev = timed_events
            if ev:
                ms, n = sum(e0.elapsed_time(e1) for _n, _k, _f, e0, e1, _t, _b in ev), len(ev)
                f, nb = sum(e[2] for e in ev), sum(e[6] for e in ev)
        ach = f / (ms * 1e-3) / 1e12
        if dom.startswith('h16'):
            # fp16-storage kernels: one fp16 MFMA per product
            peak, kname = PEAK_BF16_MFMA_TFLOPS, f'tapconv_{dom} (fp16 MFMA implicit-GEMM, fp16 storage, fp32 accumulation)'
        elif dom.startswith('wino'):
            # Winograd F(2x2,3x3) on the bf16x6 arithmetic: `achieved` counts the ALGORITHMIC FLOPs of the 3x3 convolution
            # (2 x 9 x Cin x Cout per pixel) against the same matrix-core ceiling as the direct bf16x6 kernels; the
            # kernel itself issues 16/36 of those products (+ the transforms on the VALU), reported as `executed_tflops`
            # `frac` prices them against the kernel's OWN ceiling: the dense bf16 peak / 6 partial products x 36/16 (what the
            # matrix cores would deliver in algorithmic FLOPs if the 16 executed products ran at peak) = executed bf16 MFMA FLOPs
            # over the dense bf16 peak.  The ratio to the DIRECT bf16x6 ceiling is kept as `algorithmic_vs_direct_x6_ceiling`
            peak, kname = PEAK_BF16_MFMA_TFLOPS / 6.0 * WINO_ALGORITHMIC_GAIN, (f'tapconv_{dom} (Winograd F(2x2,3x3), bf16x6-split MFMA, fp32-exact '
                                                                                'operands)')
        elif dom.startswith('x6'):
            # fp32 emulated with six bf16 MFMAs per product group: the matrix-core ceiling for algorithmic fp32 FLOPs
            # is the dense bf16 peak / 6
            peak, kname = PEAK_BF16_MFMA_TFLOPS / 6.0, f'tapconv_{dom} (bf16x6-split MFMA implicit-GEMM, fp32-exact operands)'
        else:
            peak, kname = PEAK_F32_MFMA_TFLOPS, f'tapconv_kernel<{dom}> (fp32 MFMA implicit-GEMM conv/deconv/dgrad)'
        traffic, traffic_src = pmc_traffic(dom)
        # HBM-bound kernel groups named by north_star (grid_sample, dE2000 loss, PGD step): algorithmic bytes of SURVEY 8(d)
        scale = args.batch * (args.size * args.size) / 65536.0 * 1e6
        groups = {}
        mb_group = dict(MB_PER_SCENE_256)
        if 'spaa_conv1_pair_fwd' in other:
            mb_group['warp_fwd'], mb_group['conv1_pair'] = MB_WARP_FWD_NO_CAT, MB_CONV1_PAIR
        for gname, entry in (('conv1_pair', 'spaa_conv1_pair_fwd'), ('warp_fwd', 'spaa_warp_fwd'), ('warp_fwd', 'spaa_warp_fwd_taps'),
                             ('warp_bwd_gather', 'spaa_warp_bwd_gather'),
                             ('warp_bwd_gather', 'spaa_warp_bwd_tiled'),   # (the LDS-staged form of the same adjoint)
                             ('warp_bwd_gather', 'spaa_warp_bwd_tiled_sumsq'),   # (... with ||g||^2 in its epilogue: same algorithmic bytes)
                             ('stealth_loss', 'spaa_stealth_loss_fwd_bwd'), ('step_and_track', 'spaa_step_and_track'),
                             ('step_and_track', 'spaa_step_and_track_n')):
            if entry in other:
                us = other[entry][0] * 1e3 / other[entry][1]
                gb = mb_group[gname] * scale
                groups[gname] = {'bound': 'hbm', 'achieved': round(gb / (us * 1e-6) / 1e9, 1), 'peak': PEAK_HBM_TBS * 1e3,
                                 'unit': 'GB/s', 'frac': round(gb / (us * 1e-6) / 1e12 / PEAK_HBM_TBS, 4),
                                 'avg_launch_us': round(us, 2), 'algorithmic_bytes_per_launch': round(gb)}
        other_ms = sum(v[0] for v in other.values()) / n_prof
        other_roof = sum(mb_group.values()) * scale / (PEAK_HBM_TBS * 1e12) * 1e3
        ms_step = dt / args.steps * 1e3
        roof = {'kernel': kname, 'bound': 'mfma',
                'achieved': round(ach, 2), 'peak': round(peak, 1), 'unit': 'TFLOP/s',
                'frac': round(ach / peak, 4), 'traffic': traffic, 'traffic_unit': 'bytes/launch',
                'traffic_source': f'{traffic_src} (rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE, separate passes over this '
                                  'same command; bench.py cannot collect PMC itself)',
                'algorithmic_bytes_per_launch': round(nb / n),
                'peak_note': 'algorithmic fp32 FLOP/s; dense bf16 MFMA peak 2516 TF / 6 partial products for the x6 kernels (x 36/16 for '
                             'the Winograd form, which executes 16 of the 36 products: frac = executed bf16 MFMA FLOPs / 2516 TF), '
                             '157.3 TF for the fp32-MFMA kernels',
                'avg_launch_us': round(ms * 1e3 / n, 2), 'launches_per_step': n_instr // n_prof,
                'avg_launch_us_source': 'HIP events around this kernel\'s launches inside the timed region (launch stream; every fourth step carries them)' if timed_events else 'HIP events, instrumented passes after the timed region',
                'avg_launch_us_instrumented_pass': round(ms_instr * 1e3 / n_instr, 2),
                'flop_per_launch': f / n, 'share_of_conv_time': round(ms_instr / tot_ms, 3),
                **({'executed_tflops': round(ach / WINO_ALGORITHMIC_GAIN, 2),
                    'executed_mfma_frac': round(ach / WINO_ALGORITHMIC_GAIN * 6 / PEAK_BF16_MFMA_TFLOPS, 4),
                    'executed_mfma_note': 'bf16 MFMA FLOPs the kernel issues (16 of the 36 products of a 3x3 tap set, six bf16 '
                                          'MFMAs each) over the dense bf16 peak: equal to `frac` by construction',
                    'algorithmic_vs_direct_x6_ceiling': round(ach / (PEAK_BF16_MFMA_TFLOPS / 6.0), 4)}
                   if dom.startswith('wino') else {}),
                'all_tapconv_tflops': round(sum(v[0] for v in per_tile.values()) / (tot_ms * 1e-3) / 1e12, 2),
                'conv_ms_per_step': round(tot_ms / n_prof, 3),
                'other_kernels_ms_per_step': round(other_ms, 3),
                'groups': groups,
                'pcnet_dE_hbm': pcnet_de_hbm(per_layer, other, n_prof, args.batch, args.size, args.dtype == 'f16s') if args.attack == 'spaa' else None,
                'step': {'T_roof_ms': round(t_roof_x6 * 1e3 + other_roof, 3),
                         'T_roof_ms_f32_mfma_peak': round(t_roof_f32 * 1e3 + other_roof, 3),
                         'ms_per_step': round(ms_step, 3),
                         'frac': round((t_roof_x6 * 1e3 + other_roof) / ms_step, 4),
                         'note': 'T_roof = sum over launches of max(algorithmic bytes / 8 TB/s, FLOP / peak) (SURVEY 8d); '
                                 + ('conv peak 2516 TF (fp16 storage: one fp16 MFMA per product)' if args.dtype == 'f16s' else
                                    'conv peak 419.3 TF (direct bf16x6 launches), 943.5 TF (Winograd bf16x6 launches: 36/16 of it)')
                                 + ' resp. 157.3 TF (fp32 MFMA)'}}
        table = {k: {'tile': v[3], 'gflop_per_launch': v[0] / v[2] / 1e9, 'us_per_launch': v[1] * 1e3 / v[2],
                     'tflops': v[0] / (v[1] * 1e-3) / 1e12, 'algorithmic_tb_s': v[4] / (v[1] * 1e-3) / 1e12}
                 for k, v in per_layer.items()}
        if args.profile_out:
            with open(args.profile_out, 'w') as fh:
                json.dump({'per_tile': {k: {'flop': v[0], 'ms': v[1], 'launches': v[2], 'algorithmic_bytes': v[3]}
                                        for k, v in per_tile.items()},
                           'per_layer': table,
                           'other_entry_points_us': {k: v[0] * 1e3 / v[1] for k, v in other.items()}}, fh, indent=1)

    if rank == 0:
        value = world * args.steps / dt
        out = {
            'metric': 'attack-iterations/sec (PCNet+classifier fwd/bwd), 256x256 batch=64',
            'value': round(value, 3), 'unit': 'attack-iterations/s', 'n_gpus': world, 'steps': args.steps,
            'warmup': args.warmup, 'ms_per_step': round(dt / args.steps * 1e3, 3), 'higher_is_better': True,
            'scaling': 'weak', 'vs_baseline': None,
            'dtype': 'f32' if args.dtype == 'f32' else 'f16 storage (activations/gradients), f32 accumulation, images and dE2000',
            'data': 'synthetic',
            'config': {'workload': f'{dict(resnet18="configs[1]", inception_v3="configs[2]", vgg16="configs[4] classifier, SPAA loop")[args.classifier]}: batch={args.batch} ({8} scenes x {args.batch // 8} targets) '
                                   f'{args.size}x{args.size}, {args.classifier}, '
                                   f'{"camdE_caml2" if args.attack == "spaa" else "PerC-AL adversary_projector loop body"}, per GPU',
                       'global_batch': args.batch * world, 'parallelism': f'dp{world} (independent shards)'},
            'scene_iterations_per_s': round(value * args.batch, 1),
            'per_rank_ms_per_step': [round(t / args.steps * 1e3, 3) for t in per_rank],
            'roofline': roof,
        }
        if gather_ms is not None:
            out['gather_ms'] = round(gather_ms, 3)
        # what the process group and the runtime saw (the driver can check that RCCL really ran N ranks on N devices)
        out['ranks_seen'] = {'world_size': dist.get_world_size() if dist is not None else 1,
                             'backend': dist.get_backend() if dist is not None else None,
                             'device_count': torch.cuda.device_count(), 'local_rank': local_rank}
        try:   # kernel choices of the timed configuration that were not measured table entries (spaa_amd/tapconv_tune.json)
            from spaa_amd import convplan as _cp
            _tr = _cp.tune_report()
            out['tune'] = {'borrowed_from_nearest_pixel_count': len(_tr['borrowed']), 'rule_chosen': len(_tr['untuned']),
                           'shapes': (_tr['borrowed'] + _tr['untuned'])[:12],
                           'note': 'layer-shape keys Cin_Cout_taps_sin_sout_M of this process, once-per-attack set-up launches included'}
        except Exception:   # noqa: BLE001
            pass
        if world == 1 and not args.no_cpu_baseline and args.classifier == 'resnet18' and args.dtype == 'f32' and args.attack == 'spaa':
            log(f'cpu baseline on {usable_cores()} cores')
            try:
                out['cpu_baseline'] = cpu_baseline(sd, csd, setup, scenes)
            except Exception as e:   # noqa: BLE001  (the GPU measurement above stands on its own)
                log(f'cpu baseline failed: {type(e).__name__}: {e}')
                out['cpu_baseline'] = {'error': f'{type(e).__name__}: {e}'}
        else:
            out['cpu_baseline'] = None
        default_run = world == 1 and args.classifier == 'resnet18' and args.dtype == 'f32' and args.attack == 'spaa'
        if default_run and not args.no_modes:
            # the other configurations of BASELINE.json, each timed like the headline after it (never part of `value`)
            log('extra modes: f16 storage, Inception-v3 (f32, f16 storage), VGG-16 + PerC-AL in f16 storage')
            del st
            torch.cuda.empty_cache()
            # (a failure in an extra mode must not lose the headline that has already been measured: it is recorded instead)
            def guarded(fn, *a):
                try:
                    return fn(*a)
                except Exception as e:   # noqa: BLE001
                    log(f'extra mode failed: {type(e).__name__}: {e}')
                    torch.cuda.empty_cache()
                    return {'error': f'{type(e).__name__}: {e}'}
            out['configs0_gpu'] = guarded(configs0_gpu, sd, csd, setup, scenes, dev)
            out['modes'] = {
                'configs[1] in f16 storage (resnet18, SPAA loop)': guarded(time_mode, dev, args, 'resnet18', 'f16', 'spaa'),
                'configs[2] (inception_v3 at 299x299, SPAA loop, f32)': guarded(time_mode, dev, args, 'inception_v3', 'f32', 'spaa'),
                'configs[2] in f16 storage (inception_v3 at 299x299, SPAA loop)': guarded(time_mode, dev, args, 'inception_v3', 'f16', 'spaa'),
                'configs[4] per GPU (vgg16, PerC-AL loop body, f16 storage)': guarded(time_mode, dev, args, 'vgg16', 'f16', 'perc_al'),
            }
            # the iteration counts BASELINE.json names, run back to back (the headline above is 20 steps = 0.16 s)
            out['sustained'] = {
                'configs[1]: 200 iterations (resnet18, SPAA loop, f32)': guarded(time_sustained, dev, args, 'resnet18', 'f32', 'spaa', 200),
                'configs[4] per GPU: 400 iterations (vgg16, PerC-AL loop body, f16 storage)': guarded(time_sustained, dev, args, 'vgg16', 'f16', 'perc_al', 400),
            }
        json_out.write(json.dumps(out) + '\n')
        json_out.flush()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
