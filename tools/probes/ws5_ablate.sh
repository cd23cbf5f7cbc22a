# variants of upconv_fwd_ws5_kernel (tools/build_variant.sh ws5_<X> conv_ws5.hip "-DSTJ_WS5_<X>")
cd $GRAFT_REPO_ROOT
export STJ_AB_WS5=1
for i in 1 2; do
echo ws2; STJ_AB_WS5=0 python tools/bench_conv.py --only fwd --layer 3 --iters 20
echo ws5; python tools/bench_conv.py --only fwd --layer 3 --iters 20
for v in $VARIANTS; do echo $v; STJ_LIB_PATH=strajnet_amd/variants/lib_ws5_$v.so python tools/bench_conv.py --only fwd --layer 3 --iters 20; done
done
