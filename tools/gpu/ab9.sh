cd $GRAFT_REPO_ROOT
bash tools/gpu/ab_lib.sh gs dl
MPC_LIB_PATH=$GRAFT_REPO_ROOT/rl-mpc-locomotion_amd/csrc/variants/libmpc_batch_sub6.so python tools/section_profile.py 2>&1 | tail -2
