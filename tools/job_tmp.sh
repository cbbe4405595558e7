cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/pfb3
for mib in 100000 96 48 192; do echo "== piece $mib MiB"; CHAN=0 BBT_PFB_PIECE_MIB=$mib timeout -k 10 400 python3 -u tools/bench_many_streams.py 2>&1 | grep "S=   16\|S=  128\|S= 2048"; done | tee gpurun_out/pfb3/log.txt
timeout -k 10 600 python3 -u -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "filter_bank" 2>&1 | tail -3
