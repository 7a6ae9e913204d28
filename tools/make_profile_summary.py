"""Turn rocprofv3 CSV output (copied back under gpurun_out/) into the summaries committed
under profiles/.  Usage:
  python tools/make_profile_summary.py TAG STATS_DIR FETCH_DIR WRITE_DIR BENCH_JSON "state note"
writes profiles/TAG_bench_kernel_stats.md, profiles/TAG_hbm_traffic.json, profiles/TAG_bench.json.
FETCH_SIZE / WRITE_SIZE are in KiB units; FETCH_SIZE is doubled on gfx950 (MI355X_MICROARCH.md,
HBM/rocprofv3 section; calibrated here on the CG update kernel whose traffic is known)."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

tag, stats_dir, fetch_dir, write_dir, bench_json, note = sys.argv[1:7]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def one(pattern):
    g = glob.glob(pattern, recursive=True)
    if not g:
        raise SystemExit("missing " + pattern)
    return g[0]


rows = list(csv.DictReader(open(one(os.path.join(stats_dir, '**/*kernel_stats.csv')))))
bench = open(bench_json).read().strip().splitlines()[-1]
bj = json.loads(bench)
with open(os.path.join(root, 'profiles', tag + '_bench.json'), 'w') as f:
    f.write(bench + '\n')
lines = [f"# {tag}: `rocprofv3 --kernel-trace --stats -- python bench.py --no-cpu --configs none ...` ({bj['config']['workload']})", "", note, "",
         f"bench line of the un-profiled default run (`profiles/{tag}_bench.json`): {bj['value']:.1f} {bj['unit']}, "
         f"{bj['ms_per_step']:.3f} ms/step, roofline.frac {bj['roofline']['frac']:.3f}.", "",
         "| kernel | calls | total ms | avg us | % |", "|---|---|---|---|---|"]
for r in rows[:14]:
    lines.append(f"| `{r['Name'][:100]}` | {r['Calls']} | {float(r['TotalDurationNs'])/1e6:.3f} | "
                 f"{float(r['AverageNs'])/1e3:.1f} | {r['Percentage']} |")
lines += ["", "HBM traffic per launch from separate `--pmc FETCH_SIZE` / `--pmc WRITE_SIZE` passes: "
          f"`profiles/{tag}_hbm_traffic.json`."]
open(os.path.join(root, 'profiles', tag + '_bench_kernel_stats.md'), 'w').write('\n'.join(lines) + '\n')


def pmc(d, name):
    acc = defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(one(os.path.join(d, '**/*counter_collection.csv')))):
        if r['Counter_Name'] != name:
            continue
        k = r['Kernel_Name']
        if 'pfb::' not in k:
            continue
        short = k.split('pfb::')[1].split('<')[0].split('(')[0]
        acc[short][0] += float(r['Counter_Value']) * 1024.0
        acc[short][1] += 1
    return acc


fe, wr = pmc(fetch_dir, 'FETCH_SIZE'), pmc(write_dir, 'WRITE_SIZE')
out = {}
conv = 0.0
for k in sorted(fe):
    n = fe[k][1]
    f = fe[k][0] / n
    w = wr[k][0] / max(wr[k][1], 1)
    out[k] = dict(fetch_bytes_raw=f, fetch_bytes_x2=2 * f, write_bytes=w, launches=n)
    if k.startswith(('k_row_fwd', 'k_col', 'k_row_inv')):
        conv += 2 * f + w
out['conv_group_hbm_bytes_per_launch'] = conv
# which kernels these counters saw: the commit of the tree and the hash of the kernel sources (bench.py only derives a
# rate from this file while its own kernel sources still hash to the same value)
import subprocess
sys.path.insert(0, root)
import bench as _bench
cfg = bj.get('config', {})
try:
    size = int(str(cfg.get('workload', '4096x')).split('x')[0])
except ValueError:
    size = int(os.environ.get('PFB_PROF_SIZE', 4096))
out['config'] = dict(workload='pcg', size=size, bands=int(cfg.get('bands_per_gpu', 8)), dtype=bj.get('dtype', 'f32'))
try:
    out['commit'] = subprocess.run(['git', '-C', root, 'rev-parse', '--short', 'HEAD'], capture_output=True, text=True).stdout.strip()
    out['tree_dirty'] = bool(subprocess.run(['git', '-C', root, 'status', '--porcelain', '--', 'pfb_clean_amd/csrc'],
                                            capture_output=True, text=True).stdout.strip())
except Exception:
    out['commit'] = None
out['kernel_src_sha16'] = _bench.kernel_src_hash('pcg')
json.dump(out, open(os.path.join(root, 'profiles', tag + '_hbm_traffic.json'), 'w'), indent=1)
print(open(os.path.join(root, 'profiles', tag + '_bench_kernel_stats.md')).read())
print(json.dumps({k: v for k, v in out.items() if k.startswith(('k_row', 'k_col', 'conv', 'k_pcg'))}, indent=1))
