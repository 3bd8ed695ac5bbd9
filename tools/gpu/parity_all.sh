cd $GRAFT_REPO_ROOT
python tools/parity_sweep.py 0,1,2,3,4,5,6,7,8,9,10,11,12,13,14,15,16,17,18,19,20,21,22,23,24 16 > gpurun_out/${TAG:-r04}_parity_all.txt 2>&1
grep PARITY_JSON gpurun_out/${TAG:-r04}_parity_all.txt | sed 's/^PARITY_JSON //' > gpurun_out/${TAG:-r04}_parity_sweep_seeds0_24.json
tail -c 600 gpurun_out/${TAG:-r04}_parity_all.txt
