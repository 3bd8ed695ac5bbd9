#!/bin/bash
# rocprofv3 PMC passes for the solve kernel (run on the GPU box from the repo root):
#   separate --pmc runs (TCC: FETCH_SIZE costs 3 slots, WRITE_SIZE 2 -- they do not fit one pass),
#   --kernel-trace only (no sys/hip traces together with --pmc).
# Output: gpurun_out/${PMC_TAG}_<name>.csv (per-dispatch counter rows)
set -u
export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-$PWD}
PMC_TAG=${PMC_TAG:-pmc}          # output prefix: gpurun_out/${PMC_TAG}_<pass>.csv
PMC_FLAGS=${PMC_FLAGS:-}          # extra bench.py flags (e.g. "--config 4 --robots 4096")
CMD="python $ROOT/bench.py --steps 5 --warmup 2 --repeats 1 --no-cpu-baseline --no-control-loop --no-secondary $PMC_FLAGS"
python -c "import sys; sys.path.insert(0, '$ROOT'); import rl_mpc_locomotion_amd; from rl_mpc_locomotion_amd import _lib; print(_lib.kernel_source_hash())" > $ROOT/gpurun_out/${PMC_TAG}_source.sha256
cd /tmp
for spec in "fetch:FETCH_SIZE" "write:WRITE_SIZE" "sq1:SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU" "sq2:SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_ANY SQ_WAVES"; do
  name=${spec%%:*}; ctrs=${spec#*:}
  rm -rf /tmp/pmc_$name
  rocprofv3 --pmc $ctrs --kernel-trace --output-format csv -d /tmp/pmc_$name -o p -- $CMD > $ROOT/gpurun_out/${PMC_TAG}_$name.log 2>&1
  f=$(find /tmp/pmc_$name -name "*counter_collection.csv" | head -1)
  if [ -n "$f" ]; then cp "$f" $ROOT/gpurun_out/${PMC_TAG}_$name.csv; else echo "no counter file for $name"; tail -5 $ROOT/gpurun_out/${PMC_TAG}_$name.log; fi
done
ls -la $ROOT/gpurun_out/${PMC_TAG}_*.csv
