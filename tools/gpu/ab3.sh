cd $GRAFT_REPO_ROOT
MPC_LIB_PATH=rl-mpc-locomotion_amd/csrc/variants/libmpc_batch_sub6.so MPC_SOLVE_JOBS=0 python tools/section_profile.py 4096 10 > gpurun_out/r3_sections_sub6.txt; tail -2 gpurun_out/r3_sections_sub6.txt
