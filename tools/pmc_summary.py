"""Summarise rocprofv3 --pmc CSVs: per kernel name, mean counter value per dispatch. usage: pmc_summary.py dir [dir ...]"""
import csv, glob, sys, collections, json
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for d in sys.argv[1:]:
    for fn in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(fn)):
            acc[r['Kernel_Name'].split('(')[0]][r['Counter_Name']].append(float(r['Counter_Value']))
out = {}
for k in sorted(acc):
    out[k] = {c: sum(v) / len(v) for c, v in acc[k].items()}
    out[k]['dispatches'] = max(len(v) for v in acc[k].values())
print(json.dumps(out, indent=1))
