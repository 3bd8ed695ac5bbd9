#!/bin/bash
# Full measurement pass on the GPU box (run from the repo root through gpurun); everything lands in gpurun_out/:
#   bench_n1.json            python bench.py (default flags: the driver's N=1 line)
#   kernel_stats.csv         rocprofv3 --kernel-trace --stats of python bench.py --steps 10 (kernel summary)
#   pmc_*.csv                tools/pmc_passes.sh (separate --pmc passes)
#   parity_sweep.json        tools/parity_sweep.py (HIP solver vs the oracle, seeds 0..4)
set -u
export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-$PWD}
mkdir -p $ROOT/gpurun_out
cd $ROOT
python bench.py > gpurun_out/bench_n1.json 2> gpurun_out/bench_n1.err; tail -c 600 gpurun_out/bench_n1.json
cd /tmp && rm -rf /tmp/kstats
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kstats -o k -- python $ROOT/bench.py --steps 10 --no-cpu-baseline --no-control-loop --no-secondary > $ROOT/gpurun_out/kstats.log 2>&1
f=$(find /tmp/kstats -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $ROOT/gpurun_out/kernel_stats.csv && head -4 $ROOT/gpurun_out/kernel_stats.csv | cut -c1-200
cd $ROOT && bash tools/pmc_passes.sh > gpurun_out/pmc_passes.log 2>&1; tail -4 gpurun_out/pmc_passes.log
cd $ROOT && timeout 900 python tools/parity_sweep.py > gpurun_out/parity_sweep.txt 2>&1; grep PARITY_JSON gpurun_out/parity_sweep.txt | sed 's/^PARITY_JSON //' > gpurun_out/parity_sweep.json; cut -c1-300 gpurun_out/parity_sweep.json
