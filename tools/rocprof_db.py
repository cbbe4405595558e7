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
    if m:
        return m.group(1)
    m = re.match(r'(k_\w+)', name)           # (the run-time compiled kernels are extern "C")
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


PASS_OF = (('k_osm_col256<true', 'osm_col_forward'), ('k_osm_col16<true', 'osm_col_forward'),
           ('k_osm_col4096<true', 'osm_col_forward'), ('k_osm_mid16<true', 'osm_col_forward'),
           ('k_fir_blocks', 'osm_col_forward'), ('k_osm_rowpass', 'osm_rowpass'),
           ('k_osm_col256<false', 'osm_col_inverse'), ('k_osm_col16<false', 'osm_col_inverse'),
           ('k_osm_col4096<false', 'osm_col_inverse'), ('k_osm_mid16<false', 'osm_col_inverse'))


def traffic(fetch_db, write_db, workload, tag, out_json, latest='profiles/traffic_latest.json'):
    """HBM-side bytes per dispatch from separate FETCH_SIZE / WRITE_SIZE passes
    (MI355X_MICROARCH.md, HBM section: the counters are in KiB; on gfx950
    FETCH_SIZE reports half the bytes of wide coalesced reads, so it is doubled;
    WRITE_SIZE is exact for 16-byte-per-lane stores; Infinity-Cache hits are
    counted).  Writes the per-kernel table to ``out_json`` and the per-pass sums
    (one dispatch of each kernel of a pass per chunk launch) into
    profiles/traffic_latest.json under ``workload``."""
    import os

    def per_kernel(db, counter):
        acc = defaultdict(lambda: [0.0, set()])
        con = sqlite3.connect(db)
        for name, cname, disp, value in con.execute(
                'select kernel_name, counter_name, dispatch_id, value from counters_collection'):
            s = short(name)
            if s and cname == counter:
                acc[s][0] += value
                acc[s][1].add(disp)
        return {k: (v[0] / len(v[1]), len(v[1])) for k, v in acc.items()}

    f, w = per_kernel(fetch_db, 'FETCH_SIZE'), per_kernel(write_db, 'WRITE_SIZE')
    table, passes = {}, defaultdict(float)
    for k in sorted(set(f) | set(w)):
        rd = 2.0 * f.get(k, (0.0, 0))[0] * 1024
        wr = w.get(k, (0.0, 0))[0] * 1024
        table[k] = dict(fetch_size_kib_raw=f.get(k, (0.0, 0))[0], write_size_kib=w.get(k, (0.0, 0))[0],
                        hbm_read_bytes_per_dispatch=rd, hbm_write_bytes_per_dispatch=wr,
                        dispatches=max(f.get(k, (0, 0))[1], w.get(k, (0, 0))[1]))
        for prefix, pass_name in PASS_OF:
            if k.startswith(prefix):
                passes[pass_name] += rd + wr
    json.dump(dict(workload=workload, kernels=table, passes=dict(passes),
                   note='per-dispatch averages; FETCH_SIZE x 2 (gfx950) + WRITE_SIZE, KiB -> bytes; '
                        'Infinity-Cache hits are counted'), open(out_json, 'w'), indent=1)
    all_ = json.load(open(latest)) if os.path.exists(latest) else {}
    if 'osm_rowpass' in all_:                      # round-1 layout: headline numbers at top level
        all_ = dict(headline={k: v for k, v in all_.items() if k.startswith('osm_')})
    all_[workload] = dict(passes)
    all_['source'] = f'profiles/{tag}_*_traffic.json (tools/rocprof_db.py traffic)'
    json.dump(all_, open(latest, 'w'), indent=1)
    print(json.dumps(dict(workload=workload, passes=dict(passes)), indent=1))


if __name__ == '__main__':
    if sys.argv[1] == 'stats':
        stats(sys.argv[2], sys.argv[3])
    elif sys.argv[1] == 'traffic':
        traffic(*sys.argv[2:7])
    else:
        pmc(sys.argv[2:-1], sys.argv[-1])
