cd $GRAFT_REPO_ROOT
for i in 1 2; do
for L in 2 3; do
echo base; python tools/bench_conv.py --layer $L --iters 20 --only wgrad
echo noflush; STJ_LIB_PATH=strajnet_amd/variants/lib_tr4_noflush.so python tools/bench_conv.py --layer $L --iters 20 --only wgrad
done; done
