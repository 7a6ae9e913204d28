#!/bin/bash
# tools/build_variant.sh NAME "-DFLAG=.. -DFLAG2=.." : an A/B build of the fast-path kernels with extra macros,
# linked with the other objects of the normal build into pfb_clean_amd/libpfb_hip_NAME.so (use with
# PFB_HIP_LIB=... / tools/ab_conv.py).  Prints the register / scratch use of the kernels matching $3 (a grep pattern).
set -e
cd "$(dirname "$0")/../pfb_clean_amd/csrc"
NAME=$1; FLAGS=$2; PAT=${3:-NONE}
CXX="/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -Wno-unused-variable -fno-slp-vectorize $FLAGS"
T=/tmp/pfb_variant_$NAME; mkdir -p $T
$CXX -DPFB_POW2_PART=1 -Rpass-analysis=kernel-resource-usage -c fftconv_pow2.hip -o $T/p1.o 2> $T/p1.log &
$CXX -DPFB_POW2_PART=2 -mllvm -amdgpu-sched-strategy=max-ilp -Rpass-analysis=kernel-resource-usage -c fftconv_pow2.hip -o $T/p2.o 2> $T/p2.log &
wait
make -s fftconv.o cgvec.o wavelet.o clark.o comm.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC fftconv.o cgvec.o wavelet.o clark.o comm.o $T/p1.o $T/p2.o -o ../libpfb_hip_$NAME.so
python3 - "$T" "$PAT" <<'PY'
import re, sys
t, pat = sys.argv[1], sys.argv[2]
for f in ('p1.log', 'p2.log'):
    txt = open(f'{t}/{f}').read()
    for m in re.finditer(r'Function Name: (\S+).*?VGPRs: (\d+).*?ScratchSize \[bytes/lane\]: (\d+).*?Occupancy \[waves/SIMD\]: (\d+).*?LDS Size \[bytes/block\]: (\d+)', txt, re.S):
        name = m.group(1)
        if re.search(pat, name):
            print(f'{name[:110]}  vgpr {m.group(2)} scratch {m.group(3)} occ {m.group(4)}')
PY
