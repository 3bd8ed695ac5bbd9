cd $GRAFT_REPO_ROOT
MPC_LIB_PATH=$GRAFT_REPO_ROOT/rl-mpc-locomotion_amd/csrc/variants/libmpc_batch_sub7.so python tools/section_profile.py 4096 10 exact 2>&1 | tail -3
