cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/g2s
TUNE_SPF=837284 TUNE_MIN=100 timeout -k 10 900 python3 -u tools/tune_split.py 1000 3 2>&1 | grep -v amdgpu.ids | tee gpurun_out/g2s/tune1049760.txt
