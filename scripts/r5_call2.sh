# GPU box: the hg38-scale .ufi of the reference binary against the product's two builders (scripts/r5_ufi_fullscale.py), and, while the
# reference builds on one host thread, the fault-injection check of the full-scale module (a library whose search kernel cuts slot numbers to 32 bits)
mkdir -p gpurun_out/r5b
python scripts/r5_ufi_fullscale.py --mbp 3100 --out gpurun_out/r5_ufi --ref-validate > gpurun_out/r5b/ufi_fullscale.log 2>&1 &
UFI=$!
sleep 20
( URMAPX_LIB=$PWD/urmap_amd/csrc/build_fault/liburmapx.so python -m pytest tests/test_gpu_fullscale.py -q -m gpu 2>&1 | tail -40 ) > gpurun_out/r5b/fault_slot32.txt 2>&1
wait $UFI
echo "ufi script rc=$?" >> gpurun_out/r5b/ufi_fullscale.log
tail -c 3000 gpurun_out/r5b/ufi_fullscale.log
tail -5 gpurun_out/r5b/fault_slot32.txt
