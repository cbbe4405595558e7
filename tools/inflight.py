"""How many kernels of a rocprofv3 kernel trace were in flight at once, and per-kernel statistics (dev
tool; any workload):  python tools/inflight.py <run_results.db> [last_ms]"""
import re
import sqlite3
import sys
from collections import defaultdict

db = sys.argv[1]
last_ms = float(sys.argv[2]) if len(sys.argv) > 2 else 50.
con = sqlite3.connect(db)
rows = con.execute('select S.display_name, K.stream_id, K.start, K.end from rocpd_kernel_dispatch K '
                   'join rocpd_info_kernel_symbol S on S.id=K.kernel_id and S.guid=K.guid order by K.start').fetchall()
rows = [(re.sub(r'\(.*', '', n.replace('bbt::', ''))[:60], s, a, b) for n, s, a, b in rows]
t_end = max(r[3] for r in rows)
sel = [r for r in rows if r[2] > t_end - last_ms * 1e6]
events = sorted([(r[2], 1) for r in sel] + [(r[3], -1) for r in sel])
level, last, acc = 0, events[0][0], defaultdict(float)
for t, d in events:
    acc[level] += t - last
    level += d
    last = t
total = sum(acc.values())
for k in sorted(acc):
    print(f'{k} kernels in flight: {acc[k] / 1e3:10.1f} us ({100 * acc[k] / total:5.1f} %)')
per = defaultdict(list)
for n, s, a, b in sel:
    per[(n, s)].append((b - a) / 1e3)
for (n, s), v in sorted(per.items(), key=lambda kv: -sum(kv[1])):
    print(f'stream {s} {n:60s} n={len(v):5d} mean {sum(v) / len(v):9.2f} us total {sum(v) / 1e3:8.2f} ms')
