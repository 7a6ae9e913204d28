"""Per-iteration timeline of a PCG solve from tools/trace_gaps.sh's trace.csv: kernel durations and the idle gaps
between consecutive kernels of the steady-state iterations (the 5-kernel pattern fwd, col, inv, sums, update)."""
import csv
import statistics
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
names = [r['name'] for r in rows]
st = [int(r['start_ns']) for r in rows]
en = [int(r['end_ns']) for r in rows]
# steady state: runs of the update kernel
idx = [i for i, n in enumerate(names) if 'k_pcg_update_dir' in n]
per = {}
gaps = {}
iters = []
for a, b in zip(idx[:-1], idx[1:]):
    if b - a != 5:
        continue
    iters.append((st[b] - st[a]) / 1e3)
    for k in range(a + 1, b + 1):
        key = names[k].replace('void pfb::', '')[:40]
        per.setdefault(key, []).append((en[k] - st[k]) / 1e3)
        gaps.setdefault(key, []).append((st[k] - en[k - 1]) / 1e3)
print(f"{len(iters)} steady iterations, median {statistics.median(iters):.2f} us per iteration")
print("| kernel | median duration us | median gap before it us |")
print("|---|---|---|")
td = tg = 0
for k in per:
    d, g = statistics.median(per[k]), statistics.median(gaps[k])
    td += d; tg += g
    print(f"| `{k}` | {d:.2f} | {g:.2f} |")
print(f"| sum | {td:.2f} | {tg:.2f} |")
