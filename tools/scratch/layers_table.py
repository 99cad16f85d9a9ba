"""Print bench.py --profile-out's per-layer table sorted by time."""
import json, sys
d = json.load(open(sys.argv[1]))
rows = d['layers'] if isinstance(d, dict) and 'layers' in d else d
if isinstance(rows, dict):
    rows = list(rows.values())
tot = 0
out = []
for r in rows:
    out.append(r)
for r in sorted(out, key=lambda r: -r.get('us', r.get('ms', 0))):
    print({k: (round(v, 1) if isinstance(v, float) else v) for k, v in r.items()})
