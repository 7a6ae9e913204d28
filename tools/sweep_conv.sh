cd $GRAFT_REPO_ROOT
for dt in f32 f64; do
for n in 512 1024 2048 4096; do for nb in 1 2 4 8; do
  if [ $dt = f64 ] && [ $n = 4096 ] && [ $nb = 8 ]; then continue; fi
  timeout -k 10 120 python tools/ab_conv.py --size $n --bands $nb --dtype $dt --rounds 1 base: 2>&1 | grep "^| base" | sed "s/^| base/| $dt $n x $nb/"
done; done; done
