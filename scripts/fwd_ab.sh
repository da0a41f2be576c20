for v in mlp_base mlp_a mlp_b mlp_base mlp_a mlp_b; do echo -n "$v: "; WD_HSACO_DIR=$GRAFT_REPO_ROOT/build/variants/$v timeout 60 python scripts/policy_forward_pmc.py 2>&1 | grep bf16x3; done
