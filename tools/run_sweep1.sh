mkdir -p gpurun_out/r02_sweep2
S=tools/bin/c2_ring_sweep
O=gpurun_out/r02_sweep2
timeout 300 $S 26 1000000 16 > $O/c2.txt 2>&1
timeout 300 $S 26 1000000 16 1 > $O/c2_distinct.txt 2>&1
timeout 300 $S 26 10000 16 > $O/c2_rows10k.txt 2>&1
timeout 300 $S 26 4000 16 > $O/c2_rows4k.txt 2>&1
timeout 300 $S 40 1000000 32 > $O/d32_f40.txt 2>&1
timeout 300 $S 5 10000000 64 > $O/d64_f5.txt 2>&1
tail -n 40 $O/c2.txt
