# GPU box, round 6: parity fuzz of the code that ships on an 800 Mbp index -- 17 regimes (read lengths 64 .. 1000, single reads and pairs, 1 .. 4 % substitutions), 400 k reads each, every field and path against the oracle
mkdir -p gpurun_out/r6fz
timeout 2400 python scripts/fuzz_gpu.py 800 400000 > gpurun_out/r6fz/fuzz_800mbp.txt 2>&1
tail -22 gpurun_out/r6fz/fuzz_800mbp.txt | cut -c1-160
