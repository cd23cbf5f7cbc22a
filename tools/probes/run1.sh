for i in 1 2; do
for v in base xpad16 xpad24; do
  if [ $v = base ]; then L=$PWD/strajnet_amd/libstrajnet_hip.so; else L=$PWD/strajnet_amd/variants/lib_$v.so; fi
  for l in 2 3; do echo -n "$v layer$l "; STJ_LIB_PATH=$L python tools/bench_conv.py --layer $l --only wgrad --iters 20 2>/dev/null | tail -1; done
done; done
