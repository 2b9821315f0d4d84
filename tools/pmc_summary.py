"""Summarise rocprofv3 --pmc CSV output: per kernel, average counter value per dispatch."""
import csv, glob, sys, collections
root = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(root + '/**/*counter_collection.csv', recursive=True):
    for row in csv.DictReader(open(f)):
        import re
        m = re.search(r'(\w+_kernel(<[^>]*>)?)', row['Kernel_Name'])
        k = m.group(1) if m else row['Kernel_Name'][:40]
        acc[k][row['Counter_Name']].append(float(row['Counter_Value']))
for k, d in acc.items():
    if 'at::native' in k or 'rocclr' in k:
        continue
    print(k)
    for c, v in sorted(d.items()):
        print('   %-32s n=%d avg=%.6g' % (c, len(v), sum(v) / len(v)))
