cd $GRAFT_REPO_ROOT
python bench.py 2>/dev/null > gpurun_out/bench_now.json; python - <<'PY'
import json
d=json.load(open('gpurun_out/bench_now.json'))
print('value', round(d['value']), 'ms', d['ms_per_step'], 'roofline', {k:d['roofline'][k] for k in ('kernel_ms','prep_kernel_ms','frac')})
print('exact', d['secondary'].get('exact'))
print({k:(round(v['value']) if isinstance(v,dict) and 'value' in v else v) for k,v in d['secondary'].items() if k!='exact'})
print('control', d.get('control_loop'), d.get('control_loop_with_resets'))
PY
