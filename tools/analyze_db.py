"""Per-kernel time breakdown of the LAST optimiser step (between the last two adam_ema_kernel launches) -- or of
everything -- from a rocprofv3 --kernel-trace results.db.   python tools/analyze_db.py <dir-or-db> [marker]"""
import collections
import glob
import os
import sqlite3
import sys

path = sys.argv[1]
if os.path.isdir(path):
    path = glob.glob(os.path.join(path, '**', '*results.db'), recursive=True)[0]
marker = sys.argv[2] if len(sys.argv) > 2 else 'adam'
cur = sqlite3.connect(path).cursor()
rows = list(cur.execute('select name, start, end, grid_x, grid_y, workgroup_x from kernels order by start'))
idx = [i for i, r in enumerate(rows) if marker in r[0]]
seg = rows[idx[-2] + 1:idx[-1] + 1] if len(idx) >= 2 else rows
span = (seg[-1][2] - seg[0][1]) / 1e6
busy = sum(r[2] - r[1] for r in seg) / 1e6
print(f'launches {len(seg)}  span {span:.3f} ms  busy {busy:.3f} ms')
d = collections.defaultdict(list)
for r in seg:
    n = r[0]
    n = n[n.find('trebi::') + 7:][:44] if 'trebi::' in n else n[:44]
    d[n].append(r[2] - r[1])
for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1])):
    print(f'{sum(v) / 1e4 / busy:5.1f}% n={len(v):4d} total={sum(v) / 1e6:7.3f}ms avg={sum(v) / len(v) / 1e3:8.2f}us  {k}')
