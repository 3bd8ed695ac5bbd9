cd $GRAFT_REPO_ROOT
python tools/exact_rate.py 2>&1 | tail -2
MPC_LIB_PATH=rl-mpc-locomotion_amd/csrc/variants/libmpc_batch_sub7.so python tools/section_profile.py 4096 10 exact 2>&1 | tail -1
python -m pytest tests/test_dropin.py -m gpu -x -q 2>&1 | tail -2
