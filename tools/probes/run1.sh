for c in 256 512 1024 2048; do echo "cap=$c"; STJ_LN_BWD_BLOCKS=$c python tools/bench_ln.py 2>/dev/null | grep bwd; done
