cd $GRAFT_REPO_ROOT
python tools/parity_sweep.py 25,26,27,28,29,30,31,32,33,34,35,36,37,38,39,40,41,42,43,44,45,46,47,48,49,50,51,52,53,54,55,56,57,58,59 16 > gpurun_out/r03_parity_25_59.txt 2>&1
grep PARITY_JSON gpurun_out/r03_parity_25_59.txt | sed 's/^PARITY_JSON //' > gpurun_out/r03_parity_sweep_seeds25_59.json
tail -c 400 gpurun_out/r03_parity_25_59.txt
