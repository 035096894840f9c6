cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3b
timeout 900 python -m pytest tests/test_gpu_modules.py -m gpu -x -v -k "input_grad or grad or cosine_codebook_transform" > gpurun_out/r3b/log2.txt 2>&1
grep -n "PASSED\|FAILED\|ERROR\|Fatal\|fault\|Memory" gpurun_out/r3b/log2.txt | tail -40
grep -n "Fatal Python error" -B5 -A12 gpurun_out/r3b/log2.txt | head -60
