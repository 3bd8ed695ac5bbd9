# Same-box A/B of the exact mode: ab_exact.sh <variant> <variant> ...  (tools/build_variant.sh; two alternating rounds of tools/gpu/exact_rounds.py)
cd $GRAFT_REPO_ROOT
for round in 1 2; do for v in "$@"; do echo "== $v"; MPC_LIB_PATH=$GRAFT_REPO_ROOT/rl-mpc-locomotion_amd/csrc/variants/libmpc_batch_$v.so python tools/gpu/exact_rounds.py 2>&1 | grep step; done; done
