for d in 0 16 0 16; do echo -n "dbg=$d "; STJ_WS2_DBG=$d python tools/bench_conv.py --layer 3 --only dgradE --iters 20 2>/dev/null | tail -1; done
for d in 0 16 0 16; do echo -n "fwd dbg=$d "; STJ_WS2_DBG=$d python tools/bench_conv.py --layer 3 --only fwd --iters 20 2>/dev/null | tail -1; done
