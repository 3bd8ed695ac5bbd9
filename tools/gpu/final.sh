# The round's measurement pass: GPU tests, tools/measure_round.sh full (bench line, kernel traces, PMC passes for h = 10 / 16 / 20, parity sweep seeds 0-4), the exact-mode sweep.
cd $GRAFT_REPO_ROOT
TAG=${TAG:-r06}
python -m pytest tests -m gpu -x -q 2>&1 | tail -2
TAG=$TAG bash tools/measure_round.sh full 2>&1 | tail -30
timeout 600 python tools/exact_sweep.py 2>&1 | grep "^EXACT_JSON" | sed 's/^EXACT_JSON //' > gpurun_out/${TAG}_exact_mode_sweep.json; cut -c1-300 gpurun_out/${TAG}_exact_mode_sweep.json
