#!/bin/bash
# Everything the round's artifacts come from, in one GPU session.   usage: tools/final_artifacts.sh <tag>     -> gpurun_out/<tag>_*
tag=$1
cd $GRAFT_REPO_ROOT 2>/dev/null || true
mkdir -p gpurun_out
python -m pytest tests -q -m gpu 2>&1 | tail -3 > gpurun_out/${tag}_gputests.txt
python bench.py > gpurun_out/${tag}_bench_line.json 2> gpurun_out/${tag}_bench.err
python bench.py --cfg512 --no-cpu-baseline --no-extra-configs > gpurun_out/${tag}_bench_line_cfg512.json 2>> gpurun_out/${tag}_bench.err
python bench.py --infer --steps 20 --warmup 5 > gpurun_out/${tag}_bench_line_infer.json 2>> gpurun_out/${tag}_bench.err
python bench.py --dtype f32 --no-cpu-baseline --no-extra-configs --steps 5 --warmup 2 > gpurun_out/${tag}_bench_line_f32.json 2>> gpurun_out/${tag}_bench.err
bash tools/trace.sh $tag > /dev/null 2>&1
bash tools/trace_infer.sh $tag > /dev/null 2>&1
bash tools/pmc_step.sh $tag > /dev/null 2>&1
bash tools/pmc_step.sh $tag --infer > /dev/null 2>&1
bash tools/pmc_step.sh $tag --cfg512 > /dev/null 2>&1
bash tools/trace.sh ${tag}_cfg512 --cfg512 > /dev/null 2>&1
cat gpurun_out/${tag}_gputests.txt
for f in bench_line bench_line_cfg512 bench_line_infer bench_line_f32; do python -c "import json,sys; d=json.loads(open('gpurun_out/${tag}_$f.json').read().strip().splitlines()[-1]); print('$f', d['value'], d['ms_per_step'])"; done
