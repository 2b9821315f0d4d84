"""Text summary of rocprofv3's rocpd (SQLite) output.

    python tools/rocpd_summary.py stats   <results.db>     per-kernel calls / total / average duration
    python tools/rocpd_summary.py counters <results.db>..  per-kernel average counter value per dispatch
"""
import collections
import re
import sqlite3
import sys


def short(name):
    m = re.search(r'(\w+_kernel(<[^>]*>)?)', name)
    return m.group(1) if m and 'at::native' not in name else name[:60]


def stats(path):
    c = sqlite3.connect(path)
    print('%-48s %6s %14s %14s %7s' % ('kernel', 'calls', 'total_us', 'average_us', '%'))
    for name, calls, total, avg, pct in c.execute('select name, total_calls, total_duration, average, percentage '
                                                  'from top_kernels'):
        print('%-48s %6d %14.1f %14.1f %7.2f' % (short(name), calls, total, avg, pct))


def counters(paths):
    for path in paths:
        c = sqlite3.connect(path)
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        for name, counter, value in c.execute('select kernel_name, counter_name, value from counters_collection'):
            if 'at::native' in name or 'rocclr' in name:
                continue
            acc[short(name)][counter].append(value)
        print('# ' + path)
        for k, d in acc.items():
            print(k)
            for cn, v in sorted(d.items()):
                print('   %-32s n=%d avg=%.6g' % (cn, len(v), sum(v) / len(v)))


if __name__ == '__main__':
    if sys.argv[1] == 'stats':
        stats(sys.argv[2])
    else:
        counters(sys.argv[2:])
