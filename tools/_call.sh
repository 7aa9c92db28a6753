cd /tmp; export TMPDIR=/tmp
timeout 1200 python -m pytest $GRAFT_REPO_ROOT/tests -q -m gpu -x -k "attention or attn or mha or gpt" 2>&1 | tail -4
cd $GRAFT_REPO_ROOT
python tools/kbench.py --only attn --iters 40 2>&1 | grep attn
python tools/attn_prof.py --kernel fwd 2>&1 | grep -A2 "level 0\|level 3\|us per call"
