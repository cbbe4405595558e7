"""Where the wall time of a chain of tasks goes (dev tool): kernel trace of any command
(rocprofv3 --kernel-trace, rocpd database) -> for the last busy stretches of the run (the timed
reads: stretches are separated by idle gaps of more than `gap_ms`), per kernel name the summed
duration, and how long 0, 1, 2, ... kernels were in flight at once.  Unlike tools/timeline.py it
makes no assumption about kernel names (the kernels compiled at plan time are plain `k_row`,
`k_first`, `k_last`).

    python tools/chain_timeline.py <run_results.db> [<stretches>=3] [<gap_ms>=2]
"""
import re
import sqlite3
import sys
from collections import defaultdict


def short(name):
    m = re.search(r'(?:bbt::)?(k_\w+(?:<[^>]*>)?)', name)
    return m.group(1)[:60] if m else re.sub(r'\(.*', '', name)[:60]


def main(db, stretches=3, gap_ms=2.0):
    con = sqlite3.connect(db)
    rows = con.execute('select S.display_name, K.stream_id, K.start, K.end '
                       'from rocpd_kernel_dispatch K join rocpd_info_kernel_symbol S '
                       'on S.id = K.kernel_id and S.guid = K.guid order by K.start').fetchall()
    rows = [(short(n), s, a, b) for n, s, a, b in rows]
    if not rows:
        print('no kernels in', db)
        return
    segs, start, end_max = [], 0, rows[0][3]
    for i, r in enumerate(rows):
        if r[2] - end_max > gap_ms * 1e6:
            segs.append((start, i))
            start = i
        end_max = max(end_max, r[3])
    segs.append((start, len(rows)))
    # the timed reads are the longest stretches; take the last `stretches` of those within 2x of the longest
    longest = max(rows[b - 1][3] - rows[a][2] for a, b in segs)
    big = [(a, b) for a, b in segs if max(r[3] for r in rows[a:b]) - rows[a][2] > longest / 2][-stretches:]
    for a, b in big:
        sel = rows[a:b]
        t0, t1 = sel[0][2], max(r[3] for r in sel)
        wall = (t1 - t0) / 1e3
        print(f'--- stretch of {len(sel)} kernels, {wall:.0f} us, {len(set(r[1] for r in sel))} streams')
        per = defaultdict(lambda: [0, 0.0])
        for n, _, s, e in sel:
            per[n][0] += 1
            per[n][1] += (e - s) / 1e3
        for n, (c, t) in sorted(per.items(), key=lambda kv: -kv[1][1]):
            print(f'  {n:60s} n={c:5d}  {t:10.0f} us  {100 * t / wall:5.1f} % of wall  {t / c:8.1f} us each')
        ev = sorted([(s, 1) for _, _, s, e in sel] + [(e, -1) for _, _, s, e in sel])
        depth, last, hist = 0, t0, defaultdict(float)
        for t, d in ev:
            hist[depth] += (t - last) / 1e3
            depth, last = depth + d, t
        print('  kernels in flight: ' + ', '.join(f'{k}: {100 * v / wall:.1f} %' for k, v in sorted(hist.items()) if v > 0))


if __name__ == '__main__':
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 3, float(sys.argv[3]) if len(sys.argv) > 3 else 2.0)
