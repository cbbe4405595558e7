cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/g2u; export G2_NO_OLD=1
for per in 4 16; do
for spec in "chan 3000 32768 10" "chan 1000 65536 10" "chan 6561 16384 10"; do echo "== $spec loop $per"; G2_LOOP=$per timeout -k 10 100 baseband-tasks_amd/lib/gen2_bench $spec 2>&1 | grep "us \|rtc k_loop\|FAIL"; done; done | tee gpurun_out/g2u/harness.txt
