# The round's measurement pass: GPU tests, tools/measure_round.sh full, exact-mode sweep, schedule model data.
cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -x -q 2>&1 | tail -2
bash tools/measure_round.sh full 2>&1 | tail -30
timeout 900 python tools/exact_sweep.py 2>&1 | grep "^EXACT_JSON" | sed 's/^EXACT_JSON //' > gpurun_out/r03_exact_mode_sweep.json; cut -c1-300 gpurun_out/r03_exact_mode_sweep.json
python tools/dump_schedule_data.py 4096 24 gpurun_out/sched_data.npz 2>&1 | tail -1
