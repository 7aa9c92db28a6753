timeout 900 python -m pytest tests -q -m gpu -x -k "conv2d_igemm_random" 2>&1 | tail -15
