cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -x -q 2>&1 | tail -5 > gpurun_out/r3_pytest_a.txt; cat gpurun_out/r3_pytest_a.txt
python bench.py --no-secondary --no-cpu-baseline --no-control-loop > gpurun_out/r3_bench_jobs.json 2>gpurun_out/r3_bench_jobs.err; python -c "
import json; d=json.load(open('gpurun_out/r3_bench_jobs.json')); print('JOBS', d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['prep_kernel_ms'])"
MPC_SOLVE_JOBS=0 python bench.py --no-secondary --no-cpu-baseline --no-control-loop > gpurun_out/r3_bench_nojobs.json 2>gpurun_out/r3_bench_nojobs.err; python -c "
import json; d=json.load(open('gpurun_out/r3_bench_nojobs.json')); print('NOJOBS', d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['prep_kernel_ms'])"
for c in 4 5; do python bench.py --config $c --robots 4096 --steps 5 --warmup 3 --no-secondary --no-cpu-baseline --no-control-loop 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('config', d['config']['horizon'], d['value'], d['roofline']['kernel_ms'], d['roofline']['prep_kernel_ms'])"; done
