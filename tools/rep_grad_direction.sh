# GPU box: the gradient agreement with the reference's step (tests/test_gpu_configs_640.py, configs[3]) under development
# switches: statistics epilogue on / off, with / without the specialised GEMM kernels, one stream
cd $GRAFT_REPO_ROOT
run() { echo "--- $1"; env $1 python -m pytest tests/test_gpu_configs_640.py -m gpu -q -s -k "configs3_joint_4_per_domain" 2>&1 | grep "encoder conv\|D\.[pms]  \|passed\|failed"; }
run "CGAN_X=0"
run "CGAN_X=1"
run "CGAN_FUSE_BN_STATS=0"
run "CGAN_DEBUG_GEMM_WS=9"
run "CGAN_OVERLAP=0"
