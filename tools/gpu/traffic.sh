cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -x -q 2>&1 | tail -2
python bench.py --no-secondary --no-cpu-baseline --no-control-loop 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('bench', round(d['value']), round(d['ms_per_step'],4), round(d['roofline']['kernel_ms'],4), round(d['roofline']['prep_kernel_ms'],4))"
PMC_TAG=t_pmc_h10 PMC_FLAGS="--config 2 --robots 4096" bash tools/pmc_passes.sh > gpurun_out/t_pmc.log 2>&1; tail -2 gpurun_out/t_pmc.log
