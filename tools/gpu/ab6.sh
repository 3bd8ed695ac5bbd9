cd $GRAFT_REPO_ROOT
for w in "" 1; do for c in 4 5; do env ${w:+MPC_SOLVE_JOBS_WIDE=1} python bench.py --config $c --robots 4096 --steps 5 --warmup 3 --no-secondary --no-cpu-baseline --no-control-loop 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('wide=$w config $c h', d['config']['horizon'], round(d['value']), round(d['roofline']['kernel_ms'],4), round(d['roofline']['prep_kernel_ms'],4), d['solved_fraction'])"; done; done
