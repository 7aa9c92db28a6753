cd $GRAFT_REPO_ROOT
timeout 900 python tools/soak_c1.py 60000 10000 2>&1 | tail -8
