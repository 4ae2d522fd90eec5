# GPU box, round 6 call 1: files past 4 GiB through the file-to-file path (tests/test_gpu_bigfiles.py), the same module against a library whose
# writer cuts its offset to 32 bits (must fail), BASELINE config 4's input at its own size on one device (scripts/r6_config4.py), and the
# default bench line on the deterministic genome
mkdir -p gpurun_out/r6a
{ free -g; df -h /dev/shm /tmp; nproc; cat /sys/fs/cgroup/memory.max 2>/dev/null; cat /sys/fs/cgroup/cpu.max 2>/dev/null; } > gpurun_out/r6a/box.txt 2>&1
( time python -m pytest tests/test_gpu_bigfiles.py -x -q -m gpu ) > gpurun_out/r6a/bigfiles_tests.txt 2>&1
tail -5 gpurun_out/r6a/bigfiles_tests.txt
( URMAPX_LIB=$PWD/urmap_amd/csrc/build_fault/liburmapx.so python -m pytest tests/test_gpu_bigfiles.py -q -m gpu -k "one_file" 2>&1 | tail -30 ) > gpurun_out/r6a/fault_off32.txt 2>&1
tail -4 gpurun_out/r6a/fault_off32.txt
( time URMAP_CONFIG4_CLI=1 python scripts/r6_config4.py --out gpurun_out/r6a/config4_one_device.json ) > gpurun_out/r6a/config4.log 2>&1
tail -c 1500 gpurun_out/r6a/config4.log
( time python bench.py ) > gpurun_out/r6a/bench_default.json 2> gpurun_out/r6a/bench_default.err
tail -3 gpurun_out/r6a/bench_default.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r6a/bench_default.json').read().strip().splitlines()[0])
c=d['config']
print(d['value'], d['ms_per_step'], d['parity']['bit_identical_to_oracle'], c['genome_checksum'], c['slot_table_checksum'], c['inputs_are_the_recorded_ones'], c['index_validation']['used_slots'], c['index_validation']['rows'])
for k,v in d['other_workloads'].items(): print(k, v['value'], v['ms_per_step'], v['parity']['bit_identical_to_oracle'])
e=d['e2e']
print('e2e', e['value'], e['sam_slices_checked'], 'null', e['null_sink']['value'], 'sharded', e['sharded']['value'], 'gz', {k:v['value'] for k,v in e['gz'].items()}, 'pairs', e['pairs']['value'], e['pairs']['sam_slices_checked'])
print([ (k['kernel'][:20], k['avg_ms']) for k in d['kernels']])
PY
