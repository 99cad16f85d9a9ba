"""Aggregate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes into HBM bytes per launch, per kernel.

    python tools/pmc_traffic.py --fetch <dir of pass 1> --write <dir of pass 2> --out profiles/r01_pmc_traffic.json

The two counters are collected in SEPARATE passes (FETCH_SIZE costs 3 of the 4 TCC slots, WRITE_SIZE 2:
MI355X_MICROARCH.md "rocprofv3 PMC slots").  Units and corrections as that guide prescribes:
  * both counters are reported in KiB (rocprofv3 -L: "kilobytes");
  * on gfx950 FETCH_SIZE tallies the 128-B requests of wide coalesced reads at 64 B -> doubled;
  * WRITE_SIZE is exact for 16-B-per-lane stores.
"""
import argparse
import csv
import glob
import json
import os
import re


def collect(root, counter):
    per = {}
    for path in glob.glob(os.path.join(root, '**', '*counter_collection.csv'), recursive=True):
        with open(path) as fh:
            for row in csv.DictReader(fh):
                if row['Counter_Name'] != counter:
                    continue
                name = row['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '')
                name = re.sub(r'\(.*$', '', name)
                a = per.setdefault(name.strip(), {})
                a[row['Dispatch_Id']] = a.get(row['Dispatch_Id'], 0.0) + float(row['Counter_Value'])
    return per


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--fetch', required=True)
    ap.add_argument('--write', required=True)
    ap.add_argument('--out', required=True)
    ap.add_argument('--unit-bytes', type=float, default=1024.0)
    a = ap.parse_args()
    f, w = collect(a.fetch, 'FETCH_SIZE'), collect(a.write, 'WRITE_SIZE')
    out = {}
    for name in sorted(set(f) | set(w)):
        e = {}
        if name in f:
            v = list(f[name].values())
            e['launches'] = len(v)
            e['fetch_bytes_per_launch'] = 2.0 * a.unit_bytes * sum(v) / len(v)  # gfx950: x2
        if name in w:
            v = list(w[name].values())
            e.setdefault('launches', len(v))
            e['write_bytes_per_launch'] = a.unit_bytes * sum(v) / len(v)
        e['hbm_bytes_per_launch'] = e.get('fetch_bytes_per_launch', 0.0) + e.get('write_bytes_per_launch', 0.0)
        out[name] = e
    with open(a.out, 'w') as fh:
        json.dump({'note': 'FETCH_SIZE x2 (gfx950 correction) + WRITE_SIZE, KiB -> bytes, averaged per launch; '
                           'separate --pmc passes over `bench.py --steps 3 --warmup 1 --no-cpu-baseline`',
                   'kernels': out}, fh, indent=1)
    for k, e in sorted(out.items(), key=lambda kv: -kv[1]['hbm_bytes_per_launch'] * kv[1]['launches'])[:12]:
        print(f"{k[:60]:60s} n={e['launches']:5d} fetch={e.get('fetch_bytes_per_launch', 0) / 1e6:9.2f} MB "
              f"write={e.get('write_bytes_per_launch', 0) / 1e6:9.2f} MB")


if __name__ == '__main__':
    main()
