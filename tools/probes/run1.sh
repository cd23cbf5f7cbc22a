run() { python bench.py --no-cpu-baseline --no-kernel-timing $* 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"; }
timeout 1500 python -m pytest tests -x -q -m gpu -k "loss or train or graph or step or parity" 2>&1 | tail -3
for i in 1 2 3; do echo -n "new "; run; done
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
