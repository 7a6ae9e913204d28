#!/bin/bash
# Register / scratch use of the kernels the BASELINE configs launch (hipcc -Rpass-analysis=kernel-resource-usage on the two
# objects of fftconv_pow2.hip and on cgvec.hip / wavelet.hip); exits non-zero if one of them uses scratch.
#   tools/check_scratch.sh [output.md]
set -e
cd "$(dirname "$0")/../pfb_clean_amd/csrc"
CXX="/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -Wno-unused-variable -Rpass-analysis=kernel-resource-usage"
T=/tmp/pfb_kres; mkdir -p $T
$CXX -fno-slp-vectorize -DPFB_POW2_PART=1 -c fftconv_pow2.hip -o $T/p1.o 2> $T/p1.log &
$CXX -fno-slp-vectorize -DPFB_POW2_PART=2 -mllvm -amdgpu-sched-strategy=max-ilp -c fftconv_pow2.hip -o $T/p2.o 2> $T/p2.log &
$CXX -c cgvec.hip -o $T/cg.o 2> $T/cg.log &
$CXX -mllvm -amdgpu-sched-strategy=max-ilp -c wavelet.hip -o $T/wv.o 2> $T/wv.log &
wait
cd ../..
PAT='k_row_fwd_pow2q<float, 2048, false>|k_col_pow2p<float, 4096, 8, true, true, true>|k_row_inv_pow2p<float, 2048, 8, 2, false, true>|k_pcg_update_dir<(float|double), ., 2, true, true>|k_row_fwd_pow2<float, (512|1024), (8|16)>|k_col_pow2<float, 1024, 4>|k_row_inv_pow2<float, 512, 8>|k_col_pow2p<float, 2048, 8, false, true, (true|false)>|k_row_inv_pow2p<float, 1024, 8, 0, false, true>|k_row_fwd_pow2<double, 4096, 8>|k_col_pow2x<double, 8192, 8, true, true>|k_row_inv_pow2p<double, 4096, 16, 2, false, true>|k_col_pow2x<float, 8192, 8, true, true>|k_row_fwd_pow2q<float, 4096, false>|k_row_inv_pow2p<float, 4096, 16, 2, false, true>|k_dual_update_vec<float, 4, true>|k_dwt_l1_fused<float|k_idwt_finest_fused2<float|k_pd_primal_vec<float, 4>|k_dwt_batched<float|k_idwt_batched2<float'
{
  echo "# kernel resources of the BASELINE configs' kernels (\`tools/check_scratch.sh\`: hipcc -Rpass-analysis=kernel-resource-usage)"
  echo; echo '```'
  for f in p1 p2 cg wv; do python3 tools/kres.py $T/$f.log "$PAT"; done
  echo '```'
} > ${1:-/dev/stdout}
bad=$(for f in p1 p2 cg wv; do python3 tools/kres.py $T/$f.log "$PAT"; done | awk '{ if ($(NF-2) != 0) print }')
if [ -n "$bad" ]; then echo "kernels with scratch:"; echo "$bad"; exit 1; fi
echo "no scratch in the listed kernels"
