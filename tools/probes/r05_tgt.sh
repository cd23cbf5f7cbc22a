cd $GRAFT_REPO_ROOT
echo base1024; for l in 0 1; do python tools/bench_conv.py --layer $l --only wgrad --iters 20 2>&1 | grep wgrad; done
for t in 256 512 768 1536 2048; do echo tgt$t; for l in 0 1; do STJ_LIB_PATH=strajnet_amd/variants/lib_tgt$t.so python tools/bench_conv.py --layer $l --only wgrad --iters 20 2>&1 | grep wgrad; done; done
