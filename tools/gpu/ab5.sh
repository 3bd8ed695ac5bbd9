cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -x -q 2>&1 | tail -3
python bench.py --no-secondary --no-cpu-baseline --no-control-loop 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('JOBS', round(d['value']), round(d['ms_per_step'],4), round(d['roofline']['kernel_ms'],4), round(d['roofline']['prep_kernel_ms'],4))"
MPC_SOLVE_JOBS=0 python bench.py --no-secondary --no-cpu-baseline --no-control-loop 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('NOJOBS', round(d['value']), round(d['ms_per_step'],4), round(d['roofline']['kernel_ms'],4), round(d['roofline']['prep_kernel_ms'],4))"
for c in 3 4 5; do python bench.py --config $c --robots 4096 --steps 5 --warmup 3 --no-secondary --no-cpu-baseline --no-control-loop 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('config $c h', d['config']['horizon'], round(d['value']), round(d['roofline']['kernel_ms'],4), round(d['roofline']['prep_kernel_ms'],4))"; done
