cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -x -q 2>&1 | tail -4
for v in "" 0; do MPC_SOLVE_JOBS=$v python bench.py --no-secondary --no-cpu-baseline --no-control-loop 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('JOBS=$v', round(d['value']), round(d['ms_per_step'],4), round(d['roofline']['kernel_ms'],4), round(d['roofline']['prep_kernel_ms'],4))"; done
python tools/dump_schedule_data.py 4096 16 gpurun_out/sched_jobs.npz | tail -1
