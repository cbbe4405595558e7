"""Summarise rocprofv3 rocpd databases (ROCm 7.2 default output) into small
JSON / CSV files for profiles/ (dev tool).

    python tools/rocprof_db.py stats <run_results.db> <out.csv>
    python tools/rocprof_db.py pmc <run_results.db> [<more.db> ...] <out.json>

`stats`: per-kernel calls / total / average duration (the --kernel-trace
--stats table).  `pmc`: per kernel and counter, the average value per
dispatch (summed over the instances rocprofv3 reports per dispatch).
Only kernels of this library (bbt::) are kept; names are shortened to the
template head.
"""
import json
import re
import sqlite3
import sys
from collections import defaultdict


def short(name):
    m = re.search(r'bbt::(k_\w+(?:<[^>]*>)?)', name)
    return m.group(1) if m else None


def stats(db, out):
    con = sqlite3.connect(db)
    rows = con.execute('select name, total_calls, total_duration, average, percentage from top_kernels').fetchall()
    with open(out, 'w') as f:
        f.write('"Name","Calls","TotalDurationUs","AverageUs","Percentage"\n')
        for name, calls, total, avg, pct in rows:
            s = short(name) or re.sub(r'\(.*', '', name)[:60]
            f.write(f'"{s}",{calls},{total:.3f},{avg:.3f},{pct:.4f}\n')
            print(f'{s:48s} calls={calls:6d} avg={avg:10.3f} us  {pct:6.2f} %')


def pmc(dbs, out):
    acc = defaultdict(lambda: defaultdict(lambda: [0.0, set()]))
    meta = {}
    for db in dbs:
        con = sqlite3.connect(db)
        q = ('select kernel_name, counter_name, dispatch_id, value, vgpr_count, lds_block_size, '
             'grid_size, workgroup_size, duration from counters_collection')
        for name, counter, disp, value, vgpr, lds, grid, wg, dur in con.execute(q):
            s = short(name)
            if not s:
                continue
            a = acc[s][counter]
            a[0] += value
            a[1].add((db, disp))
            meta[s] = dict(vgpr=vgpr, lds=lds, grid=grid, workgroup=wg)
    res = {}
    for k, counters in acc.items():
        res[k] = dict(meta[k])
        res[k]['dispatches'] = max(len(a[1]) for a in counters.values())
        res[k]['per_dispatch'] = {c: a[0] / len(a[1]) for c, a in sorted(counters.items())}
    json.dump(res, open(out, 'w'), indent=1)
    print(json.dumps(res, indent=1))


if __name__ == '__main__':
    if sys.argv[1] == 'stats':
        stats(sys.argv[2], sys.argv[3])
    else:
        pmc(sys.argv[2:-1], sys.argv[-1])
