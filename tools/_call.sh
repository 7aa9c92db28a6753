cd /tmp; export TMPDIR=/tmp
timeout 900 python -m pytest $GRAFT_REPO_ROOT/tests -q -m gpu -x -k "cross_entropy or ce_ or loss or gpt" 2>&1 | tail -3
cd $GRAFT_REPO_ROOT
for s in 8 0 2 32; do echo "share=$s"; NNHIP_CE_SHARE=$s python tools/kbench.py --only ce --iters 60 2>&1 | grep "ce "; done
