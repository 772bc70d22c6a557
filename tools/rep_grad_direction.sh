# GPU box: the encoder's gradient agreement with the reference's step (tests/test_gpu_configs_640.py, configs[3]) under
# development switches: statistics epilogue on / off, with / without the x-resident 1x1 kernel
cd $GRAFT_REPO_ROOT
run() { echo "--- $1"; env $1 python -m pytest tests/test_gpu_configs_640.py -m gpu -q -s -k "configs3_joint_4_per_domain" 2>&1 | grep "encoder conv\|encoder bn\|passed\|failed"; }
run "CGAN_X=0"
run "CGAN_FUSE_BN_STATS=0"
run "CGAN_DEBUG_GEMM_WS=12"
run "CGAN_DEBUG_GEMM_WS=9"
run "CGAN_OVERLAP=0"
