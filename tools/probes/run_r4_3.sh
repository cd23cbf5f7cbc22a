cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_ops_gpu.py -x -q -k "inference_heads" 2>&1 | tail -6
timeout 600 python -m pytest tests/test_model_gpu.py -x -q -k "config4 or fp16_inference" 2>&1 | tail -3
for i in 1 2; do timeout 600 python bench.py --infer --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.readlines()[-1]); print('infer', d['value'], d['ms_per_step'], d.get('roofline',{}).get('kernel'), d.get('roofline',{}).get('frac')); print({k:(v['ms_per_step'],v['launches']) for k,v in list(d.get('families',{}).items())[:8]})"; done
