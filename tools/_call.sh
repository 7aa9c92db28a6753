cd /tmp; export TMPDIR=/tmp
timeout 900 python -m pytest $GRAFT_REPO_ROOT/tests -q -m gpu -x -k "deferred_weight" 2>&1 | tail -8
