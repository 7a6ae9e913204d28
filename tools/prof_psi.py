"""Development aid: runs psi.dot / psi.hdot of config #4 a few times (for rocprofv3 --kernel-trace) and, with
--parse DIR, prints the mean duration per (kernel, grid) from the trace CSV."""
import sys, glob, csv, collections
if len(sys.argv) > 2 and sys.argv[1] == '--parse':
    f = glob.glob(sys.argv[2] + '/**/*kernel_trace.csv', recursive=True)[0]
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        g = (r.get('Grid_Size_X') or r.get('Grid_Size'), r.get('Grid_Size_Y'), r.get('Grid_Size_Z'))
        acc[(r['Kernel_Name'][:60], g)].append(int(r['End_Timestamp']) - int(r['Start_Timestamp']))
    for k, v in sorted(acc.items(), key=lambda kv: -sum(kv[1])):
        v = v[len(v) // 3:]
        print(f"{k[0]:60s} grid {k[1]}  n={len(v):3d}  mean {sum(v)/len(v)/1e3:8.1f} us")
    sys.exit(0)
import torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pfb_clean_amd.operators.psi import Psi
n, nb = 2048, 4
dt = torch.float64 if (len(sys.argv) > 1 and sys.argv[1] == 'f64') else torch.float32
bases = ['self', 'db1', 'db2', 'db3', 'db4']
psi = Psi(nb, n, n, bases, 3, 1, dtype=dt)
x = torch.randn((nb, n, n), dtype=dt, device='cuda')
a = torch.zeros((nb, len(bases), psi.Nymax, psi.Nxmax), dtype=dt, device='cuda')
out = torch.empty_like(x)
for _ in range(12):
    psi.dot(x, a)
    psi.hdot(a, out)
torch.cuda.synchronize()
