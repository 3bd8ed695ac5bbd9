# The round's measurement pass: GPU tests, tools/measure_round.sh full, the big parity sweep, the exact-mode sweep.
cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -x -q 2>&1 | tail -2
bash tools/measure_round.sh full 2>&1 | tail -30
bash tools/gpu/parity_all.sh | tail -c 300
timeout 900 python tools/exact_sweep.py 2>&1 | grep "^EXACT_JSON" | sed 's/^EXACT_JSON //' > gpurun_out/r03_exact_mode_sweep.json; cut -c1-300 gpurun_out/r03_exact_mode_sweep.json
