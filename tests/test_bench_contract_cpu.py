"""CPU: the bench line committed under profiles/ (produced by `python bench.py` on the GPU box, tools/collect_profiles.sh)
carries every field of the measurement contract, and its numbers are mutually consistent."""
import glob
import json
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def latest_bench():
    def key(f):
        m = re.search(r'r(\d+)_v(\d+)_bench', os.path.basename(f))
        return (int(m.group(1)), int(m.group(2)))

    files = sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r*_v*_bench.json')), key=key)
    assert files, 'no committed bench line under profiles/'
    with open(files[-1]) as fh:
        return json.load(fh), files[-1]


def test_bench_line_fields():
    b, path = latest_bench()
    for k in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling',
              'vs_baseline', 'dtype', 'data', 'config', 'roofline', 'cpu_baseline'):
        assert k in b, (k, path)
    assert b['higher_is_better'] is True and b['scaling'] == 'weak' and b['vs_baseline'] is None
    assert b['dtype'] == 'f32' and b['data'] == 'synthetic' and 'workload' in b['config'] and 'model' not in b['config']
    assert abs(b['value'] - b['n_gpus'] * 1e3 / b['ms_per_step']) < 0.02 * b['value']
    r = b['roofline']
    for k in ('bound', 'achieved', 'peak', 'unit', 'frac', 'traffic'):
        assert k in r, k
    assert r['bound'] in ('hbm', 'mfma') and r['unit'] in ('GB/s', 'TFLOP/s')
    assert abs(r['frac'] - r['achieved'] / r['peak']) < 2e-3
    assert abs(r['achieved'] - r['flop_per_launch'] / (r['avg_launch_us'] * 1e-6) / 1e12) < 0.02 * r['achieved']
    if r['kernel'].startswith('tapconv_wino') and int(re.search(r'r(\d+)_', os.path.basename(path)).group(1)) >= 6:
        # round 6 on: a Winograd kernel is priced against its OWN ceiling (dense bf16 peak / 6 x 36/16), i.e. `frac` is the executed
        # bf16 MFMA FLOPs over the dense bf16 peak; the ratio to the direct bf16x6 ceiling is a separate field
        assert abs(r['peak'] - 2516.0 / 6 * 36 / 16) < 0.5
        assert abs(r['frac'] - r['executed_mfma_frac']) < 2e-3
        assert abs(r['algorithmic_vs_direct_x6_ceiling'] - r['achieved'] / (2516.0 / 6)) < 2e-3
    c = b['cpu_baseline']
    for k in ('value', 'unit', 'cores', 'kind', 'sample'):
        assert k in c, k
    assert c['kind'] in ('reference', 'port') and c['cores'] >= 1 and c['value'] > 0


def test_rocprof_summary_agrees_with_bench_roofline():
    """The committed rocprofv3 --stats summary of the same command: its average duration of the dominant kernel agrees
    with the HIP-event average the bench line reports (within 5 %)."""
    import csv
    import re
    b, path = latest_bench()
    stats = path.replace('_bench.json', '_rocprofv3_kernel_stats.csv')
    assert os.path.exists(stats), stats
    kern = b['roofline']['kernel']
    if kern.startswith('tapconv_wino'):
        nm = kern.split(' ')[0]
        if 'false, 8>' in open(stats).read() or 'false, 4>' in open(stats).read():   # (round 4 on: <N tile, variant, 0, canvas, two sources, waves>)
            bn, var, nw = ((64, 2, 4) if '_8x32x64' in nm else (64, 2, 8) if 'x64' in nm.replace('_canvas', '').replace('_2src', '')[-4:] else (128, 3, 8))
            want = f"wino_x6_kernel<{bn}, {var}, 0, {'true' if '_canvas' in nm else 'false'}, {'true' if '_2src' in nm else 'false'}, {nw}>"
        else:
            want = 'wino_x6_kernel<64, ' if 'x64 ' in kern or nm.endswith('x64') else 'wino_x6_kernel<128, '
    else:
        m = re.match(r'tapconv_(x6d(?:16)?(?:co)?(?:a3)?)_(\d+)x(\d+)', kern)
        assert m, kern
        fam, bm, bn = m.groups()
        want = (f'tapconv_x6d_kernel<{int(bm) // 32}, {bn}, {16 if "16" in fam else 32}, {"true" if "co" in fam else "false"}, '
                f'{3 if fam.endswith("a3") else 2}>')
    with open(stats) as fh:
        rows = [r for r in csv.DictReader(fh) if want in r['Name']]
    assert len(rows) == 1, (want, len(rows))
    avg_us = float(rows[0]['AverageNs']) / 1e3
    assert abs(avg_us - b['roofline']['avg_launch_us']) < 0.05 * avg_us, (avg_us, b['roofline']['avg_launch_us'])


def test_gpus_flag_launches_that_many_ranks():
    """`python bench.py --gpus 2` (no torchrun environment) must start 2 ranks itself and print ONE line with n_gpus 2:
    rehearsed here on gloo with a stand-in step (`--rehearse-glue`: launcher, rendezvous, barrier-bracketed timed region,
    MAX over ranks, preallocated result gather)."""
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_PORT')}
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '3', '--warmup', '1',
                        '--rehearse-glue'], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip().startswith('{')]
    assert len(lines) == 1, r.stdout
    b = json.loads(lines[0])
    assert b['n_gpus'] == 2 and len(b['per_rank_ms_per_step']) == 2 and b['gather_ok'] is True and b['value'] is None
    assert b['steps'] == 3 and b['warmup'] == 1 and b['gather_ms'] >= 0
    # a rank count that contradicts the environment is an error, not a silent 1-rank run
    env2 = dict(env, WORLD_SIZE='1', RANK='0', LOCAL_RANK='0')
    r2 = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--rehearse-glue'],
                        capture_output=True, text=True, timeout=120, env=env2)
    assert r2.returncode != 0 and 'WORLD_SIZE=1' in r2.stderr


def test_eight_rank_rehearsal_as_the_driver_launches_it():
    """The driver's own N = 8 launch line (`python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1
    --master-port P bench.py --gpus 8 ...`), rehearsed on gloo with the stand-in step: rendezvous from RANK / WORLD_SIZE / MASTER_*,
    barrier-bracketed timed region, MAX over the 8 ranks, the preallocated gather of 8 blocks, ONE line from rank 0."""
    import socket
    import subprocess
    import sys
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_PORT', 'MASTER_ADDR')}
    env['OMP_NUM_THREADS'] = '1'
    r = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '8', '--master-addr',
                        '127.0.0.1', '--master-port', str(port), os.path.join(ROOT, 'bench.py'), '--gpus', '8', '--steps', '3',
                        '--warmup', '1', '--rehearse-glue'], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip().startswith('{')]
    assert len(lines) == 1, r.stdout
    b = json.loads(lines[0])
    assert b['n_gpus'] == 8 and len(b['per_rank_ms_per_step']) == 8 and b['gather_ok'] is True and b['value'] is None
