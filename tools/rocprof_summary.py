"""Summarise rocprofv3 output of `bench.py` into profiles/ (dev tool).

    python tools/rocprof_summary.py gpurun_out/prof r01

Reads <dir>/stats/**/_kernel_stats.csv, <dir>/pmc_fetch/**/_counter_collection.csv
and <dir>/pmc_write/**; writes profiles/<tag>_kernel_stats.csv,
profiles/<tag>_pmc_summary.json and profiles/traffic_latest.json.

HBM bytes per launch follow MI355X_MICROARCH.md (HBM section): counters are in
KiB; on gfx950 FETCH_SIZE reports half of the bytes of a wide (16 B/lane)
coalesced read stream, so it is doubled; WRITE_SIZE is exact for 16 B/lane
streaming stores.  Infinity-Cache hits are included in both.
"""
import csv
import glob
import json
import os
import shutil
import sys
from collections import defaultdict

SHORT = {'k_osm_rowpass': 'osm_rowpass', 'k_osm_col256<true': 'osm_col_forward',
         'k_osm_col256<false': 'osm_col_inverse', 'k_fft_rows': 'channelize_fft_rows',
         'k_osm_col16<true': 'osm_col_forward', 'k_osm_col16<false': 'osm_col_inverse',
         'k_osm_small': 'osm_small', 'k_pfb': 'pfb', 'k_seam_fix': 'seam_fix'}


def short(name):
    for k, v in SHORT.items():
        if k in name:
            return v
    return None


def pmc(dirname, counter):
    acc = defaultdict(lambda: [0.0, 0])
    for path in glob.glob(os.path.join(dirname, '**', '*_counter_collection.csv'), recursive=True):
        with open(path) as f:
            for row in csv.DictReader(f):
                if row['Counter_Name'] != counter:
                    continue
                s = short(row['Kernel_Name'])
                if s:
                    acc[s][0] += float(row['Counter_Value'])
                    acc[s][1] += 1
    return {k: (v[0] / v[1], v[1]) for k, v in acc.items()}


def main():
    src, tag = sys.argv[1], sys.argv[2]
    os.makedirs('profiles', exist_ok=True)
    stats = glob.glob(os.path.join(src, 'stats', '**', '*_kernel_stats.csv'), recursive=True)
    summary = {'kernels': {}}
    if stats:
        shutil.copy(stats[0], f'profiles/{tag}_kernel_stats.csv')
        with open(stats[0]) as f:
            for row in csv.DictReader(f):
                s = short(row['Name'])
                if s:
                    summary['kernels'][s] = dict(calls=int(row['Calls']),
                                                 avg_us=float(row['AverageNs']) / 1e3,
                                                 percent=float(row['Percentage']))
    fetch = pmc(os.path.join(src, 'pmc_fetch'), 'FETCH_SIZE')
    write = pmc(os.path.join(src, 'pmc_write'), 'WRITE_SIZE')
    traffic = {}
    for k in sorted(set(fetch) | set(write)):
        f_kib, nf = fetch.get(k, (0.0, 0))
        w_kib, nw = write.get(k, (0.0, 0))
        rd = 2.0 * f_kib * 1024          # gfx950 correction for wide coalesced reads
        wr = w_kib * 1024
        summary['kernels'].setdefault(k, {}).update(
            fetch_size_kib_raw=f_kib, write_size_kib=w_kib, hbm_read_bytes_per_launch=rd,
            hbm_write_bytes_per_launch=wr, pmc_launches=max(nf, nw))
        traffic[k] = rd + wr
    summary['note'] = ('per-launch averages; FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 '
                       'reports half of wide coalesced reads); Infinity-Cache hits are counted')
    json.dump(summary, open(f'profiles/{tag}_pmc_summary.json', 'w'), indent=1)
    json.dump(traffic, open('profiles/traffic_latest.json', 'w'), indent=1)
    print(json.dumps(summary, indent=1))


if __name__ == '__main__':
    main()
