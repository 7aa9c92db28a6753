O=gpurun_out/r03l; mkdir -p $O
timeout 900 python -m pytest tests -q -m gpu -x -k "conv" 2>&1 | tail -5
python tools/kbench.py --only convg,conv --iters 20 2>&1 | grep -v amdgpu | tee $O/kb_conv.txt
NNHIP_CONV_DIRECT=0 python tools/kbench.py --only conv --iters 20 2>&1 | grep -v amdgpu | tee $O/kb_conv_nodirect.txt
