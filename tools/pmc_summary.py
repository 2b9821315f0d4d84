"""Summarise rocprofv3 --pmc CSV output: per kernel, average counter value per dispatch.

    python tools/pmc_summary.py DIR [DIR ...]                       text summary
    python tools/pmc_summary.py --json OUT --units N [--fetch-scale 2] DIR [DIR ...]

With --json the summary is also written as the file bench.py reads its measured HBM traffic and instruction
counts from (profiles/pmc_traffic.json): it records the hash of the kernel sources the counters were collected
on, so that bench.py reports `traffic: null` rather than a stale figure once a kernel changes.  N is the number
of frames per launch of the profiled command (the model kernels score N - 29 windows).  HBM bytes per launch =
fetch_scale x FETCH_SIZE + WRITE_SIZE (both KiB): gfx950 tallies the 128-byte requests of coalesced
streaming reads at 64 bytes; the scale for this code's access widths is measured by
tools/probes/fetch_calibration.hip.
"""
import argparse
import collections
import csv
import glob
import json
import os
import re
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def kernel_name(raw):
    m = re.search(r'(\w+_kernel(<[^>]*>)?)', raw)
    return m.group(1) if m else raw[:40]


def collect(roots):
    """roots: directories of rocprofv3 passes (rocpd *.db or *counter_collection.csv inside) or such files."""
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for root in roots:
        csvs = [root] if root.endswith('.csv') else glob.glob(root + '/**/*counter_collection.csv', recursive=True)
        dbs = [root] if root.endswith('.db') else glob.glob(root + '/**/*.db', recursive=True)
        for f in csvs:
            for row in csv.DictReader(open(f)):
                acc[kernel_name(row['Kernel_Name'])][row['Counter_Name']].append(float(row['Counter_Value']))
        for f in dbs:
            import sqlite3
            con = sqlite3.connect(f)
            for name, counter, value in con.execute('select kernel_name, counter_name, value from counters_collection'):
                acc[kernel_name(name)][counter].append(float(value))
    return {k: {c: (len(v), sum(v) / len(v)) for c, v in d.items()} for k, d in acc.items()
            if 'at::native' not in k and 'rocclr' not in k}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('roots', nargs='+')
    ap.add_argument('--json')
    ap.add_argument('--units', type=int, help='frames per launch of the profiled command')
    ap.add_argument('--fetch-scale', type=float, default=2.0)
    ap.add_argument('--command', default='')
    args = ap.parse_args()
    summary = collect(args.roots)
    for k, d in summary.items():
        print(k)
        for c, (n, avg) in sorted(d.items()):
            print('   %-32s n=%d avg=%.6g' % (c, n, avg))
    if args.json:
        from bench import csrc_hash
        kernels = {}
        for k, d in summary.items():
            rec = {c: avg for c, (n, avg) in d.items()}
            rec['units'] = args.units if k.startswith(('mfcc', 'delta', 'synth')) else args.units - 29
            if 'FETCH_SIZE' in rec and 'WRITE_SIZE' in rec:
                rec['hbm_bytes'] = (args.fetch_scale * rec['FETCH_SIZE'] + rec['WRITE_SIZE']) * 1024.0
            kernels[k] = rec
        with open(args.json, 'w') as f:
            json.dump({'csrc_hash': csrc_hash(), 'command': args.command, 'fetch_scale': args.fetch_scale,
                       'frames_per_launch': args.units, 'kernels': kernels}, f, indent=1, sort_keys=True)
        print('wrote', args.json)


if __name__ == '__main__':
    main()
