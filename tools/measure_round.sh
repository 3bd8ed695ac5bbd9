#!/bin/bash
# Full measurement pass on the GPU box (run from the repo root through gpurun); everything lands in gpurun_out/ with the prefix $TAG (default r03):
#   ${TAG}_bench_n1.json                 python bench.py (default flags: the driver's N = 1 line)
#   ${TAG}_kernel_stats_bench_steps10.csv  rocprofv3 --kernel-trace --stats of python bench.py --steps 10 --repeats 1 (kernel summary)
#   ${TAG}_kernel_stats_config{4,5}.csv  the same for the long horizons (4096 robots)
#   ${TAG}_pmc_h{10,16,20}_{fetch,write,sq1,sq2}.csv   separate --pmc passes per horizon (tools/pmc_passes.sh)
#   ${TAG}_parity_sweep.json             tools/parity_sweep.py (HIP solver vs the oracle, seeds 0..4)
# usage: bash tools/measure_round.sh [quick]      (quick: bench line + h = 10 kernel trace only)
set -u
export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-$PWD}
TAG=${TAG:-r06}
MODE=${1:-full}
mkdir -p $ROOT/gpurun_out
cd $ROOT
python bench.py > gpurun_out/${TAG}_bench_n1.json 2> gpurun_out/${TAG}_bench_n1.err; tail -c 300 gpurun_out/${TAG}_bench_n1.json; echo
kstats() {   # $1 = output name, rest = bench flags
  local name=$1; shift
  cd /tmp && rm -rf /tmp/kstats
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kstats -o k -- python $ROOT/bench.py "$@" --repeats 1 --no-cpu-baseline --no-control-loop --no-secondary > $ROOT/gpurun_out/${TAG}_kstats_$name.log 2>&1
  f=$(find /tmp/kstats -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $ROOT/gpurun_out/${TAG}_kernel_stats_$name.csv && head -4 $ROOT/gpurun_out/${TAG}_kernel_stats_$name.csv | cut -c1-220
  cd $ROOT
}
kstats bench_steps10 --steps 10
[ "$MODE" = quick ] && exit 0
kstats config4_4096xh16 --config 4 --robots 4096 --steps 5 --warmup 2
kstats config5_4096xh20 --config 5 --robots 4096 --steps 5 --warmup 2
for hc in "10:2" "16:4" "20:5"; do
  H=${hc%%:*}; C=${hc#*:}
  PMC_TAG=${TAG}_pmc_h$H PMC_FLAGS="--config $C --robots 4096" bash tools/pmc_passes.sh > gpurun_out/${TAG}_pmc_h$H.log 2>&1; tail -5 gpurun_out/${TAG}_pmc_h$H.log
done
timeout 900 python tools/parity_sweep.py > gpurun_out/${TAG}_parity_sweep.txt 2>&1; grep PARITY_JSON gpurun_out/${TAG}_parity_sweep.txt | sed 's/^PARITY_JSON //' > gpurun_out/${TAG}_parity_sweep.json; cut -c1-300 gpurun_out/${TAG}_parity_sweep.json
timeout 600 python tools/controller_parity.py > gpurun_out/${TAG}_controller_parity.txt 2>&1; grep CONTROLLER_PARITY_JSON gpurun_out/${TAG}_controller_parity.txt | sed 's/^CONTROLLER_PARITY_JSON //' > gpurun_out/${TAG}_controller_parity.json; cut -c1-300 gpurun_out/${TAG}_controller_parity.json
