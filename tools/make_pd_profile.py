"""Summarise a `rocprofv3 --kernel-trace -- python tools/bench_pd.py 2048 4` trace into
profiles/<tag>_pd_kernel_stats.md: kernel time per steady-state primal-dual iteration.
usage: python tools/make_pd_profile.py TAG TRACE_DIR [WALL_MS]"""
import csv, glob, sys, collections
tag, d = sys.argv[1], sys.argv[2]
wall = sys.argv[3] if len(sys.argv) > 3 else None
f = glob.glob(d + '/**/*kernel_trace.csv', recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
# the timed 20-iteration run is the tail of the trace: find the last 20 k_pd_primal launches
idx = [i for i, r in enumerate(rows) if 'k_pd_primal' in r['Kernel_Name']]
nit = 17
lo, hi = idx[-nit - 1], idx[-1]                  # exactly `nit` iterations between two primal updates
seg = rows[lo + 1:hi + 1]
per = collections.defaultdict(lambda: [0, 0])
for r in seg:
    name = r['Kernel_Name'].split('(')[0]
    per[name][0] += 1
    per[name][1] += int(r['End_Timestamp']) - int(r['Start_Timestamp'])
span = (int(rows[hi]['End_Timestamp']) - int(rows[lo]['End_Timestamp'])) / nit / 1e3
ktot = sum(v[1] for v in per.values()) / nit / 1e3
out = [f"# {tag}_pd: primal-dual backward step, BASELINE config #4 (2048^2 x 4 bands, self + db1..db4, 3 levels, fp32)", "",
       f"`rocprofv3 --kernel-trace -- python tools/bench_pd.py 2048 4`; {nit} steady-state iterations of `primal_dual_optimised`:",
       f"GPU span {span / 1e3:.3f} ms per iteration, kernel time {ktot / 1e3:.3f} ms per iteration, "
       f"{sum(v[0] for v in per.values()) / nit:.0f} launches per iteration" + (f"; un-profiled wall {wall} ms per iteration." if wall else "."), "",
       "| kernel | launches / iteration | us / iteration |", "|---|---|---|"]
for name, (n, t) in sorted(per.items(), key=lambda kv: -kv[1][1]):
    out.append(f"| `{name[:80]}` | {n / nit:.1f} | {t / nit / 1e3:.1f} |")
open(f'profiles/{tag}_pd_kernel_stats.md', 'w').write('\n'.join(out) + '\n')
print('\n'.join(out))
