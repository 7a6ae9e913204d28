"""Markdown table of the BASELINE configs from ONE default `bench.py` run (its "configs" object; round 3).
    python tools/make_configs_table.py BENCH_JSON [TITLE]"""
import json
import sys

d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
title = sys.argv[2] if len(sys.argv) > 2 else "the BASELINE configs from one default `python bench.py` run"
print(f"# {title}\n")
print("| config | value | ms per step (K+1 matvecs / K steps) | row_fwd / col / row_inv ms per launch | roofline.frac (algorithmic) |")
print("|---|---|---|---|---|")


def row(name, o):
    if 'error' in o:
        print(f"| {name} | error: {o['error'][:80]} | | | |")
        return
    r = o.get('roofline') or {}
    st = r.get('stage_ms')
    stages = f"{st['row_fwd']:.4f} / {st['col']:.4f} / {st['row_inv']:.4f}" if st else "(whole iteration)"
    print(f"| {name} | {o['value']:.1f} {o['unit']} | {o['ms_per_step']:.4f} ({o['steps']} steps x {o['repeats']}) | {stages} | {r.get('frac')} |")


row("C3 4096^2 x 8 fp32 (headline)", d)
names = {"C1_1024x1_f32": "C1 1024^2 x 1 fp32", "C2_4096x1_f32": "C2 4096^2 x 1 fp32 (= the 8-GPU shard of C3)",
         "C2_4096x1_f32_exchange_world1": "C2 with the native RCCL exchange live (nccl, world 1, child process)",
         "C4_pd_2048x4_f32": "C4 2048^2 x 4 primal-dual iteration (self + db1..db4, 3 levels)",
         "C5_shard_8192x2_f64": "C5 shard 8192^2 x 2 fp64"}
for k, o in d.get('configs', {}).items():
    row(names.get(k, k), o)
cb = d.get('cpu_baseline')
if cb:
    print(f"\nCPU oracle (same run): {cb['value']} {cb['unit']} on {cb['cores']} cores -> GPU / CPU = {d.get('gpu_over_cpu')}; "
          f"parity.conv_relerr {d['parity']['conv_relerr']:.2e} (tol {d['parity']['tol']}).")
