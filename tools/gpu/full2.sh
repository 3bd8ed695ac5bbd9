cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -x -q 2>&1 | tail -3
python bench.py > gpurun_out/r3_bench_n1.json 2> gpurun_out/r3_bench_n1.err; python - <<'PY'
import json
d=json.load(open('gpurun_out/r3_bench_n1.json'))
print('value', round(d['value']), 'ms', round(d['ms_per_step'],4), 'solve', round(d['roofline']['kernel_ms'],4), 'prep', round(d['roofline']['prep_kernel_ms'],4), 'frac', round(d['roofline']['frac'],4))
for k,v in d.get('secondary',{}).items(): print(k, round(v['control_steps_per_s']), {kk: vv for kk, vv in v.items() if kk not in ('what','control_steps_per_s')})
print('control_loop', d['control_loop']['robot_ticks_per_s'], 'resets', d['control_loop_with_resets']['robot_ticks_per_s'], 'err', d['max_grf_err_vs_osqp'])
PY
timeout 900 python tools/exact_sweep.py 256 2>&1 | grep -v "^EXACT_JSON" | tail -5
