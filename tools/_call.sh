cd /tmp; export TMPDIR=/tmp
timeout 2400 python -m pytest $GRAFT_REPO_ROOT/tests -q -m gpu -x -p no:cacheprovider 2>&1 | tail -2
