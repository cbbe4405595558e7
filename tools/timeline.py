"""Timeline of a rocprofv3 --kernel-trace run (rocpd database): where the wall
time of the overlap-save lanes goes (dev tool).

    python tools/timeline.py <run_results.db> [<out.json>]

For the longest busy stretch of the run (the timed steps) it reports, per
stream (= plan lane): kernel time, the gaps between one kernel's end and the
next one's start, and -- over all streams together -- how long 0, 1, 2, ...
kernels were in flight at once.
"""
import json
import re
import sqlite3
import sys
from collections import defaultdict


def short(name):
    m = re.search(r'bbt::(k_\w+(?:<[^>]*>)?)', name)
    return m.group(1) if m else re.sub(r'\(.*', '', name)[:40]


def main(db, out=None):
    con = sqlite3.connect(db)
    rows = con.execute('select S.display_name, K.stream_id, K.queue_id, K.start, K.end '
                       'from rocpd_kernel_dispatch K join rocpd_info_kernel_symbol S '
                       'on S.id = K.kernel_id and S.guid = K.guid order by K.start').fetchall()
    rows = [(short(n), s, q, a, b) for n, s, q, a, b in rows]
    # the timed region: the last contiguous stretch of bbt kernels with no idle gap > 2 ms
    osm = [r for r in rows if r[0].startswith('k_osm') or r[0].startswith('k_seam')]
    if not osm:
        print('no overlap-save kernels in', db)
        return
    seg_start = 0
    end_max = osm[0][4]
    segs = []
    for i, r in enumerate(osm):
        if r[3] - end_max > 2_000_000:
            segs.append((seg_start, i))
            seg_start = i
        end_max = max(end_max, r[4])
    segs.append((seg_start, len(osm)))
    a, b = max(segs, key=lambda s: s[1] - s[0])
    sel = osm[a:b]
    # the plan's own streams (lanes / stages): the window they are active in (bench.py's
    # isolated-pass measurement runs on the caller's stream and is left out)
    counts = defaultdict(int)
    for r in sel:
        if not r[0].startswith('k_seam'):
            counts[(r[1], r[2])] += 1
    main = min(counts) if counts else None
    lanes = [k for k in counts if k != main]
    if lanes:
        lo = min(r[3] for r in sel if (r[1], r[2]) in lanes)
        hi = max(r[4] for r in sel if (r[1], r[2]) in lanes)
        sel = [r for r in sel if r[3] >= lo and r[4] <= hi]
    t0, t1 = sel[0][3], max(r[4] for r in sel)
    wall = (t1 - t0) / 1e3
    res = dict(db=db, kernels=len(sel), wall_us=wall)
    print(f'{len(sel)} kernels over {wall:.1f} us')
    per_stream = defaultdict(list)
    for r in sel:
        per_stream[(r[1], r[2])].append(r)
    res['streams'] = {}
    for key, ks in sorted(per_stream.items()):
        busy = sum(k[4] - k[3] for k in ks) / 1e3
        gaps = defaultdict(list)
        for p, n in zip(ks[:-1], ks[1:]):
            gaps[f'{p[0].split("<")[0]}{"F" if "<true" in p[0] else ""} -> {n[0].split("<")[0]}{"F" if "<true" in n[0] else ""}'].append((n[3] - p[4]) / 1e3)
        print(f'stream {key}: {len(ks)} kernels, busy {busy:.1f} us ({100 * busy / wall:.1f} % of wall)')
        g = {}
        for name, v in sorted(gaps.items()):
            v = sorted(v)
            g[name] = dict(n=len(v), median=v[len(v) // 2], mean=sum(v) / len(v), p90=v[int(0.9 * len(v))])
            print(f'    gap {name:40s} n={len(v):5d} median {v[len(v) // 2]:7.2f} mean {sum(v) / len(v):7.2f} '
                  f'p90 {v[int(0.9 * len(v))]:7.2f} us')
        res['streams'][str(key)] = dict(kernels=len(ks), busy_us=busy, gaps=g)
    by = defaultdict(list)
    for r in sel:
        by[r[0]].append((r[4] - r[3]) / 1e3)
    res['durations'] = {}
    for name, v in sorted(by.items()):
        v = sorted(v)
        res['durations'][name] = dict(n=len(v), median=v[len(v) // 2], mean=sum(v) / len(v))
        print(f'  {name:52s} n={len(v):5d} median {v[len(v) // 2]:8.2f} mean {sum(v) / len(v):8.2f} us')
    # concurrency histogram
    ev = []
    for r in sel:
        ev.append((r[3], 1))
        ev.append((r[4], -1))
    ev.sort()
    level, last, hist = 0, t0, defaultdict(float)
    for t, d in ev:
        hist[level] += (t - last) / 1e3
        last = t
        level += d
    res['in_flight_us'] = {str(k): v for k, v in sorted(hist.items())}
    for k, v in sorted(hist.items()):
        print(f'  {k} kernels in flight: {v:10.1f} us ({100 * v / wall:5.1f} %)')
    if out:
        json.dump(res, open(out, 'w'), indent=1)


if __name__ == '__main__':
    main(*sys.argv[1:3])
