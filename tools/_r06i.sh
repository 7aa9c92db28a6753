cd $GRAFT_REPO_ROOT
O=gpurun_out/r06i; mkdir -p $O
for cfg in "1 1" "3 3"; do set -- $cfg
echo "== B=$1 H=$2"
for P in 0 1; do
NNHIP_ATTN_SB_FWD=stream CMP_PAD=$P CMP_B=$1 CMP_H=$2 python tools/attn_sb_fwd_cmp.py save /tmp/a.npz 2>&1 | grep -v amdgpu
for M in lds pw; do
NNHIP_ATTN_SB_FWD=$M CMP_PAD=$P CMP_B=$1 CMP_H=$2 python tools/attn_sb_fwd_cmp.py save /tmp/b.npz 2>&1 | grep -v amdgpu
echo "-- $M pad=$P"; python tools/attn_sb_fwd_cmp.py cmp /tmp/a.npz /tmp/b.npz | tail -1
done; done; done
for M in stream lds pw; do NNHIP_ATTN_SB_FWD=$M timeout 900 python -m pytest tests/test_hip_parity.py -x -q -m gpu -k "balanced or attention" > $O/tests_$M.log 2>&1; echo "tests $M: $(tail -1 $O/tests_$M.log)"; done
for M in stream lds pw stream lds pw; do echo -n "$M: "; NNHIP_ATTN_SB_FWD=$M timeout 300 python tools/attn_sb_time.py 2>&1 | grep -v amdgpu | tail -1; done
