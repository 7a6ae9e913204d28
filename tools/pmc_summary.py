"""Aggregate rocprofv3 --pmc counter_collection CSVs per kernel (development aid).
usage: python tools/pmc_summary.py DIR [DIR ...]   -> mean counter value per launch, per kernel family"""
import csv, glob, sys, collections, re
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for d in sys.argv[1:]:
    for f in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            name = r['Kernel_Name']
            m = re.search(r'pfb::(k_\w+)', name)
            if not m:
                continue
            acc[m.group(1)][r['Counter_Name']].append((r['Dispatch_Id'], float(r['Counter_Value'])))
for k, cs in sorted(acc.items()):
    print(k)
    for c, vals in sorted(cs.items()):
        per = collections.defaultdict(float)
        for did, v in vals:
            per[did] += v                     # sum over XCDs / SEs of one dispatch
        vs = list(per.values())
        print(f"    {c:28s} {sum(vs)/len(vs):16.1f}   ({len(vs)} launches)")
