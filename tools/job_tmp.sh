cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 500 python3 -u tools/bench_generic.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/generic_lengths.txt &&
timeout -k 10 300 python3 -u tools/bench_generic.py chan:14 chan:30 chan:360 chan:2187 chan:4374 chan:6174 2>&1 | grep Chan | tee -a gpurun_out/generic_lengths.txt
