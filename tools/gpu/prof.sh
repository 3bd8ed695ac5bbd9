# In-kernel section timers of the solve kernel: needs tools/build_variant.sh prof -DMPC_SECTION_PROFILE first.
cd $GRAFT_REPO_ROOT
MPC_LIB_PATH=$GRAFT_REPO_ROOT/rl-mpc-locomotion_amd/csrc/variants/libmpc_batch_prof.so python tools/section_profile.py 2>&1 | tail -4
