# role ablation of swin_mlp_fwd's k-step: variants built with tools/build_variant.sh abl_<X> swin_fused.hip "-DSTJ_ABL_<X>"
cd $GRAFT_REPO_ROOT
echo base; python tools/bench_swin_k.py --only mlp 2>&1 | grep "mlp fwd"
for v in NOGELU NOLDS1 NOLDS2 ALL; do echo $v; STJ_LIB_PATH=strajnet_amd/variants/lib_abl_$v.so python tools/bench_swin_k.py --only mlp 2>&1 | grep "mlp fwd"; done
