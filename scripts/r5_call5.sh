# GPU box: phase 6's rounds chosen per call (four for reads over 192 bases), chunks capped at 524 288 reads, the faster gzip decoder
mkdir -p gpurun_out/r5f
python -m pytest tests/test_gpu_parity.py tests/test_gpu_slow.py tests/test_gpu_phase3.py -x -q -m gpu > gpurun_out/r5f/parity_tests.txt 2>&1
tail -4 gpurun_out/r5f/parity_tests.txt
( time python bench.py ) > gpurun_out/r5f/bench_full.json 2> gpurun_out/r5f/bench_full.err
tail -4 gpurun_out/r5f/bench_full.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r5f/bench_full.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['parity']['bit_identical_to_oracle'], d['phase6']['launch_ms_by_round'])
e=d['e2e']
print('e2e', e['value'], e['seconds'], e['first_run_seconds'], e['bound'], 'null', e['null_sink']['value'], e['null_sink']['stream_time_s_summed_over_lanes'], 'sharded', e['sharded']['value'], e['sharded']['vs_one_file'])
print('gz', {k:(v['value'], v['inflate_GBs'], v['seconds'], v['sam_records_identical_to_plain_run']) for k,v in e['gz'].items()})
print('pairs', e['pairs']['value'], e['pairs']['sam_records_identical_to_oracle'])
for n,o in d['other_workloads'].items(): print(n, o['value'], o['ms_per_step'], o['parity']['bit_identical_to_oracle'], [(k['kernel'][:20],k['avg_ms']) for k in o['kernels']][:3], o.get('phase6',{}).get('launch_ms_by_round'), o.get('phase6',{}).get('dropped_by_a_round_gate_before_their_dp'), o.get('phase6',{}).get('hsps_given_to_dp_kernel'))
PY
