cd $GRAFT_REPO_ROOT
timeout 300 python tools/kbench.py --only headpad 2>&1 | grep -v amdgpu
