cd $GRAFT_REPO_ROOT
O=gpurun_out/r06j; mkdir -p $O
timeout 1800 python -m pytest tests/test_hip_parity.py -x -q -m gpu -k "wave_private or counted" > $O/tests.log 2>&1; tail -3 $O/tests.log
NNHIP_ATTN_SB_FWD=pw timeout 900 python -m pytest tests/test_hip_parity.py -x -q -m gpu -k "balanced or attention" > $O/tests_pw.log 2>&1; echo "tests pw: $(tail -1 $O/tests_pw.log)"
for M in stream pw; do echo -n "$M: "; NNHIP_ATTN_SB_FWD=$M timeout 300 python tools/attn_sb_time.py 2>&1 | grep -v amdgpu | tail -1; done
