# scratch script for gpurun experiments: alternate two builds / settings inside ONE call (box-to-box spread is +-1.5 %)
run() { python bench.py --no-cpu-baseline --no-kernel-timing $* 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"; }
for i in 1 2 3; do echo -n "base "; run; echo -n "variant "; STJ_LIB_PATH=$PWD/strajnet_amd/variants/lib_variant.so run; done
