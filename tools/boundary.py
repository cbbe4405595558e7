"""Kernel timeline around the call boundaries of a rocprofv3 kernel trace (dev tool):
python tools/boundary.py <run_results.db> [n_boundaries]"""
import re
import sqlite3
import sys

db = sys.argv[1]
nb = int(sys.argv[2]) if len(sys.argv) > 2 else 2
con = sqlite3.connect(db)
rows = con.execute('select S.display_name, K.stream_id, K.queue_id, K.start, K.end from rocpd_kernel_dispatch K '
                   'join rocpd_info_kernel_symbol S on S.id=K.kernel_id and S.guid=K.guid order by K.start').fetchall()


def short(n):
    m = re.search(r'bbt::(k_\w+)(<[^>]*>)?', n)
    if not m:
        return n[:30]
    s = m.group(1)
    if s == 'k_osm_col256':
        s += 'F' if m.group(2).startswith('<true') else 'L'
    return s


rows = [(short(n), s, q, a, b) for n, s, q, a, b in rows]
osm = [r for r in rows if r[0].startswith('k_osm') or r[0].startswith('k_seam')]
t_end = osm[-1][4]
sel = [r for r in osm if r[3] > t_end - 125e6]
from collections import Counter
cnt = Counter(r[1] for r in sel if r[0].startswith('k_osm'))
lanes = [k for k, _ in cnt.most_common(2)]
lane = [r for r in sel if r[1] == lanes[0]]
gaps = [(lane[i + 1][3] - lane[i][4]) / 1e3 for i in range(len(lane) - 1)]
big = [i for i, g in enumerate(gaps) if g > 40]
print('lane', lanes[0], 'gaps > 40 us at kernel index', [(i, round(gaps[i])) for i in big])
for gi in big[:nb]:
    t0 = lane[gi][4]
    print('--- boundary (lane %d last kernel ends at 0)' % lanes[0])
    for r in sel:
        if t0 - 120e3 < r[3] < t0 + 500e3:
            print(f'{r[0]:16s} stream {r[1]} start {(r[3] - t0) / 1e3:9.1f} end {(r[4] - t0) / 1e3:9.1f}')
