"""A/B of convolution-kernel variants on ONE device (guide rule 24: never rank builds across boxes).
Variants are environment settings the library reads once per process, so each measurement is a child process;
the variants are interleaved over several rounds and the median stage times reported.

    python tools/ab_conv.py [--size 4096] [--bands 8] [--dtype f32] [--rounds 3] NAME:ENV=V,ENV=V ...
e.g. python tools/ab_conv.py base: seq0:PFB_FWD_SEQ=0 spread0:PFB_SPREAD=0
"""
import argparse
import json
import os
import statistics
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r'''
import sys, json, torch
sys.path.insert(0, %(root)r)
from pfb_clean_amd import _lib, _dev
from pfb_clean_amd.operators.psf import PsfConvPlan
n, nb, f64, beam = %(n)d, %(nb)d, %(f64)d, %(beam)d
dt = torch.float64 if f64 else torch.float32
cdt = torch.complex128 if f64 else torch.complex64
dev = torch.device('cuda')
g = torch.Generator(device=dev).manual_seed(1)
psfhat = (torch.rand((nb, 2 * n, n + 1), generator=g, device=dev, dtype=dt) / nb).to(cdt)
plan = PsfConvPlan(psfhat, n, n, 2 * n)
x = torch.randn((nb, n, n), generator=g, device=dev, dtype=dt)
r = torch.randn((nb, n, n), generator=g, device=dev, dtype=dt)
b = torch.rand((nb, n, n), generator=g, device=dev, dtype=dt) if beam else None
out = torch.empty_like(x)
dots = torch.zeros(3, dtype=torch.float64, device=dev)
lib = _lib.load()
def apply():
    _lib.check(lib.pfb_psfconv_apply_dots(plan._h, 0, nb, _dev.ptr(x), _dev.ptr(b), 0.0, 0.1, _dev.ptr(out),
                                          _dev.ptr(x), _dev.ptr(r), _dev.ptr(dots), _dev.stream()))
for _ in range(5): apply()
torch.cuda.synchronize()
plan.set_profiling(1)
for _ in range(40): apply()
torch.cuda.synchronize()
ms, k = plan.get_profile()
print(json.dumps({"fwd": ms[0] / k, "col": ms[1] / k, "inv": ms[2] / k, "sum": sum(ms) / k, "check": float(out.double().abs().sum())}))
'''


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--size', type=int, default=4096)
    ap.add_argument('--bands', type=int, default=8)
    ap.add_argument('--dtype', default='f32')
    ap.add_argument('--beam', type=int, default=0)
    ap.add_argument('--rounds', type=int, default=3)
    ap.add_argument('variants', nargs='+')
    a = ap.parse_args()
    code = CHILD % dict(root=ROOT, n=a.size, nb=a.bands, f64=int(a.dtype == 'f64'), beam=a.beam)
    res = {}
    for rnd in range(a.rounds):
        for v in a.variants:
            name, _, envs = v.partition(':')
            env = dict(os.environ)
            for kv in filter(None, envs.split(',')):
                k, _, val = kv.partition('=')
                env[k] = val
            o = subprocess.run([sys.executable, '-c', code], env=env, capture_output=True, text=True, timeout=300)
            line = [ln for ln in o.stdout.splitlines() if ln.startswith('{')]
            if o.returncode != 0 or not line:
                print(f"{name}: FAILED rc={o.returncode}\n{o.stderr[-800:]}", flush=True)
                continue
            res.setdefault(name, []).append(json.loads(line[-1]))
    print(f"# A/B {a.size}^2 x {a.bands} {a.dtype} beam={a.beam}, {a.rounds} interleaved rounds, median ms per launch (min)")
    print("| variant | row_fwd | col | row_inv | sum | checksum |")
    print("|---|---|---|---|---|---|")
    for name, rs in res.items():
        def m(k):
            v = [r[k] for r in rs]
            return f"{statistics.median(v):.4f} ({min(v):.4f})"
        print(f"| {name} | {m('fwd')} | {m('col')} | {m('inv')} | {m('sum')} | {rs[0]['check']:.9e} |", flush=True)


if __name__ == '__main__':
    main()
