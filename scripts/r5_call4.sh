# GPU box: multi-rank / placement / gzip tests on the new host code, then the whole default bench line (10 M-read file-to-file leg)
mkdir -p gpurun_out/r5d
python -m pytest tests/test_gpu_multi.py tests/test_gpu_phase3.py -x -q -m gpu -s > gpurun_out/r5d/multi_tests.txt 2>&1
tail -6 gpurun_out/r5d/multi_tests.txt
python -m pytest tests/test_gpu_text.py -x -q -m gpu > gpurun_out/r5d/text_tests.txt 2>&1
tail -3 gpurun_out/r5d/text_tests.txt
( time python bench.py ) > gpurun_out/r5d/bench_full.json 2> gpurun_out/r5d/bench_full.err
tail -5 gpurun_out/r5d/bench_full.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r5d/bench_full.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['parity']['bit_identical_to_oracle'], d['cpu_baseline']['value'], d['cpu_baseline'].get('port_value'))
e=d['e2e']
print('e2e', e['value'], e['reads'], e['seconds'], e['bound'], 'null', e['null_sink']['value'], e['null_sink']['stream_time_s_summed_over_lanes'], 'sharded', e['sharded']['value'], e['sharded']['vs_one_file'])
print('gz', {k:(v['value'], v['inflate_GBs'], v['sam_records_identical_to_plain_run']) for k,v in e['gz'].items()})
print('pairs', e['pairs']['value'], e['pairs']['sam_records_identical_to_oracle'])
print('ref', e.get('reference_binary'))
print(e.get('placement'))
PY
