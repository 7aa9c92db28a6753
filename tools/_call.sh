cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_hip_parity.py -m gpu -q -k "conv2d_mfma" -x 2>&1 | tail -8
