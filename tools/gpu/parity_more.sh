# extended GRF parity sweep: seeds 5..39 over configs 2-5 (tools/parity_sweep.py; the round's measurement pass covers seeds 0..4)
cd $GRAFT_REPO_ROOT
TAG=${TAG:-r06}
python tools/parity_sweep.py $(seq -s, 5 39) 16 > gpurun_out/${TAG}_parity_5_39.txt 2>&1
grep PARITY_JSON gpurun_out/${TAG}_parity_5_39.txt | sed 's/^PARITY_JSON //' > gpurun_out/${TAG}_parity_sweep_seeds5_39.json
tail -c 600 gpurun_out/${TAG}_parity_5_39.txt
