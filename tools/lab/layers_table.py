"""Print bench.py --profile-out's per-layer table sorted by time (optionally against a second file)."""
import json, sys
def load(f):
    d = json.load(open(f))
    pl = d['per_layer']
    rows = pl if isinstance(pl, list) else [dict(name=k, **v) if 'name' not in v else v for k, v in pl.items()]
    return d, rows
d, rows = load(sys.argv[1])
old = {r['name']: r for r in load(sys.argv[2])[1]} if len(sys.argv) > 2 else {}
print('tapconv total %.0f us' % sum(r['us_per_launch'] for r in rows))
for r in sorted(rows, key=lambda r: -r['us_per_launch'])[:int(sys.argv[3]) if len(sys.argv) > 3 else 40]:
    o = old.get(r['name'])
    print(f"{r['name']:24s} {r['tile']:22s} {r['us_per_launch']:7.1f} us {r['tflops']:7.1f} TF {r['algorithmic_tb_s']:5.2f} TB/s" + (f"   was {o['us_per_launch']:7.1f} {o['tile']}" if o else ''))
o = d['other_entry_points_us']
print('other %.0f us' % sum(o.values()), {k: round(v, 1) for k, v in o.items()})
