# kernel trace of a short bench run (per-kernel average durations): gpurun_out/${TAG}_kernel_stats.csv
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
TAG=${TAG:-r05t}
mkdir -p gpurun_out
cd /tmp && rm -rf /tmp/kstats
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kstats -o k -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --repeats 1 --no-cpu-baseline --no-control-loop --no-secondary ${TRACE_FLAGS:-} > $GRAFT_REPO_ROOT/gpurun_out/${TAG}_kstats.log 2>&1
f=$(find /tmp/kstats -name "*kernel_stats.csv" | head -1); cp "$f" $GRAFT_REPO_ROOT/gpurun_out/${TAG}_kernel_stats.csv
cut -d, -f1-4 $GRAFT_REPO_ROOT/gpurun_out/${TAG}_kernel_stats.csv | sed 's/(anonymous namespace):://; s/_ZN12_GLOBAL__N_1//' | cut -c1-60,150-400 | head -12
grep "^{\"metric" $GRAFT_REPO_ROOT/gpurun_out/${TAG}_kstats.log | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('value', round(d['value']), 'ms/step', round(d['ms_per_step'],4), 'solve', round(r['kernel_ms'],4), 'prep', round(r['prep_kernel_ms'],4), 'outside', round(r['step_ms_outside_the_two_solver_kernels'],4))"
