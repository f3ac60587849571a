mkdir -p gpurun_out/r4
REPS=5 timeout -k 10 300 python scripts/x1_check.py 8192 16384 > gpurun_out/r4/x1_check_e.txt 2>&1
echo "## alt build max-ilp" >> gpurun_out/r4/x1_check_e.txt
IEACHE_LIBRARY=$PWD/ie-ache_amd/csrc/build/alt_maxilp/libieache.so REPS=5 timeout -k 10 300 python scripts/x1_check.py 8192 16384 >> gpurun_out/r4/x1_check_e.txt 2>&1
tail -30 gpurun_out/r4/x1_check_e.txt
