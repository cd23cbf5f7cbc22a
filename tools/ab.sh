run() { python bench.py --no-cpu-baseline --no-kernel-timing 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"; }
echo -n "base "; run
for t in 384 512 1024 1536; do echo -n "SPLITK_TGT=$t "; STJ_SPLITK_TGT=$t run; done
echo -n "SPLITK_CAP=48 "; STJ_SPLITK_CAP=48 run
echo -n "SPLITK_CAP=192 "; STJ_SPLITK_CAP=192 run
echo -n "NO_RS "; STJ_NO_RS=1 run
echo -n "RS_MIN_M=65536 "; STJ_RS_MIN_M=65536 run
echo -n "GEMM_SMALL=128 "; STJ_GEMM_SMALL=128 run
echo -n "GEMM_SMALL=512 "; STJ_GEMM_SMALL=512 run
echo -n "base "; run
