cd $GRAFT_REPO_ROOT
python tools/dump_schedule_data.py 4096 16 gpurun_out/sched_jobs.npz
MPC_SOLVE_JOBS=0 python tools/dump_schedule_data.py 4096 16 gpurun_out/sched_nojobs.npz
MPC_LIB_PATH=rl-mpc-locomotion_amd/csrc/variants/libmpc_batch_prof.so MPC_SOLVE_JOBS=0 python tools/section_profile.py 4096 10 > gpurun_out/r3_sections_b.txt; tail -2 gpurun_out/r3_sections_b.txt
