# GPU box: the file-to-file leg in a light bench run and in the full one -- what in the full run's history slows it?
mkdir -p gpurun_out/r5g
export URMAP_BENCH_INDEX_CACHE=/dev/shm/urmap_idx URMAP_BENCH_NO_E2E_GZ=1 URMAP_BENCH_NO_E2E_PAIRS=1
show() { python - <<PY
import json
d=json.loads(open('gpurun_out/r5g/$1.json').read().strip().splitlines()[-1])
e=d['e2e']
print('$1: e2e', round(e['value']/1e6,2), 'first', e['first_run_seconds'], 'sec', e['seconds'], e['stage_busy_s'], 'null', round(e['null_sink']['value']/1e6,2), e['null_sink']['lane_busy_s_summed'], e['null_sink']['stream_time_s_summed_over_lanes'], 'sharded', round(e['sharded']['value']/1e6,2))
PY
}
URMAP_BENCH_NO_REFERENCE=1 python bench.py --steps 2 --warmup 1 --no-other-workloads --no-cpu-baseline > gpurun_out/r5g/light.json 2> gpurun_out/r5g/light.err; show light
URMAP_BENCH_NO_REFERENCE=1 URMAPX_NO_NUMA_PIN=1 python bench.py --steps 2 --warmup 1 --no-other-workloads --no-cpu-baseline > gpurun_out/r5g/light_nopin.json 2> gpurun_out/r5g/light_nopin.err; show light_nopin
URMAP_BENCH_NO_REFERENCE=1 python bench.py --steps 2 --warmup 1 --no-other-workloads > gpurun_out/r5g/with_cpu_baseline.json 2> gpurun_out/r5g/with_cpu_baseline.err; show with_cpu_baseline
URMAP_BENCH_NO_REFERENCE=1 OMP_WAIT_POLICY=passive python bench.py --steps 2 --warmup 1 --no-other-workloads > gpurun_out/r5g/with_cpu_baseline_passive.json 2> gpurun_out/r5g/with_cpu_baseline_passive.err; show with_cpu_baseline_passive
URMAP_BENCH_NO_REFERENCE=1 python bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r5g/with_other_workloads.json 2> gpurun_out/r5g/with_other_workloads.err; show with_other_workloads
rm -rf /dev/shm/urmap_idx
