"""Register / scratch use per kernel from a hipcc -Rpass-analysis=kernel-resource-usage log.
    python tools/kres.py LOG [regex on the demangled name]"""
import re, sys, subprocess
txt = open(sys.argv[1]).read()
pat = sys.argv[2] if len(sys.argv) > 2 else '.'
rows = re.findall(r'Function Name: (\S+).*?VGPRs: (\d+).*?AGPRs: (\d+).*?ScratchSize \[bytes/lane\]: (\d+).*?Occupancy \[waves/SIMD\]: (\d+).*?LDS Size \[bytes/block\]: (\d+)', txt, re.S)
names = subprocess.run(['c++filt'], input='\n'.join(r[0] for r in rows), capture_output=True, text=True).stdout.splitlines()
seen = set()
for r, n in zip(rows, names):
    n = n.replace('void pfb::', '')
    n = n[:n.find('(')] if '(' in n else n
    if re.search(pat, n) and n not in seen:
        seen.add(n)
        print(f'{n:90s} vgpr {r[1]:>3} agpr {r[2]:>3} scratch {r[3]:>4} occ {r[4]}')
