cd $GRAFT_REPO_ROOT
timeout 600 python examples/conv_classifier.py --steps 40 2>&1 | tail -6
