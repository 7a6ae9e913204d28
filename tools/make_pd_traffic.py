"""HBM traffic of one primal-dual iteration (BASELINE config #4) from the two PMC passes of
`tools/collect_profile.sh TAG --workload pd` -> profiles/TAG_pd_hbm_traffic.json (+ the kernel-stats table as .md).
    python tools/make_pd_traffic.py TAG STATS_DIR FETCH_DIR WRITE_DIR
Per kernel: MEAN bytes per launch x launches per iteration (launch count / number of primal updates, rounded); FETCH_SIZE
and WRITE_SIZE are in KiB, FETCH_SIZE is doubled on gfx950 (MI355X_MICROARCH.md, HBM / rocprofv3 section)."""
import csv
import glob
import json
import os
import subprocess
import sys
from collections import defaultdict

tag, stats_dir, fetch_dir, write_dir = sys.argv[1:5]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
import bench as _bench          # noqa: E402


def one(pattern):
    g = glob.glob(pattern, recursive=True)
    if not g:
        raise SystemExit("missing " + pattern)
    return g[0]


def pmc(d, name):
    acc = defaultdict(lambda: [0.0, 0])
    per = defaultdict(float)
    for r in csv.DictReader(open(one(os.path.join(d, '**/*counter_collection.csv')))):
        if r['Counter_Name'] != name or 'pfb::' not in r['Kernel_Name']:
            continue
        per[(r['Kernel_Name'], r['Dispatch_Id'])] += float(r['Counter_Value']) * 1024.0
    for (k, _), v in per.items():
        short = k.split('(')[0]
        acc[short][0] += v
        acc[short][1] += 1
    return acc


fe, wr = pmc(fetch_dir, 'FETCH_SIZE'), pmc(write_dir, 'WRITE_SIZE')
nit = max(v[1] for k, v in fe.items() if 'k_pd_primal' in k)
kernels, total = {}, 0.0
for k in sorted(fe):
    per_it = round(fe[k][1] / nit)
    if per_it < 1:
        continue                                    # set-up kernels (re-layout, synthetic inputs)
    f2 = 2.0 * fe[k][0] / fe[k][1] * per_it
    w = wr[k][0] / max(wr[k][1], 1) * per_it
    kernels[k] = dict(launches_per_iteration=per_it, fetch_bytes_x2=f2, write_bytes=w)
    total += f2 + w
out = {"config": {"workload": "pd", "size": 2048, "bands": 4, "dtype": "f32"},
       "note": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) of `bench.py --workload pd --no-cpu --configs none "
               "--steps 4`; KiB units, FETCH_SIZE doubled on gfx950 (MI355X_MICROARCH.md); per kernel: MEAN over its launches x "
               "launches per iteration",
       "kernels": kernels, "hbm_bytes_per_iteration": total, "iterations_profiled": nit,
       "kernel_src_sha16": _bench.kernel_src_hash('pd')}
try:
    out['commit'] = subprocess.run(['git', '-C', root, 'rev-parse', '--short', 'HEAD'], capture_output=True, text=True).stdout.strip()
except Exception:
    out['commit'] = None
json.dump(out, open(os.path.join(root, 'profiles', tag + '_pd_hbm_traffic.json'), 'w'), indent=1)
rows = list(csv.DictReader(open(one(os.path.join(stats_dir, '**/*kernel_stats.csv')))))
lines = [f"# {tag}: `rocprofv3 --kernel-trace --stats -- python bench.py --workload pd --no-cpu --configs none --steps 20 --warmup 5 --repeats 2`",
         "", "| kernel | calls | total ms | avg us | % |", "|---|---|---|---|---|"]
for r in rows[:16]:
    lines.append(f"| `{r['Name'][:100]}` | {r['Calls']} | {float(r['TotalDurationNs'])/1e6:.3f} | {float(r['AverageNs'])/1e3:.1f} | {r['Percentage']} |")
lines += ["", f"HBM traffic per iteration (PMC passes): {total/1e9:.3f} GB -> `profiles/{tag}_pd_hbm_traffic.json`."]
open(os.path.join(root, 'profiles', tag + '_pd_kernel_stats.md'), 'w').write('\n'.join(lines) + '\n')
print('\n'.join(lines))
