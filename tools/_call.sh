cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_hip_parity.py -m gpu -q -k "upstream or cross_entropy or mse or conv_classifier" -x 2>&1 | tail -12
