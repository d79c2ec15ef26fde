#!/bin/bash
export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
OUT=gpurun_out/r4p; mkdir -p $OUT
(time timeout 1500 python bench.py --steps 20 --warmup 3 > $OUT/bench_full.json 2> $OUT/bench_full.err); tail -3 $OUT/bench_full.err
python - <<'PY'
import json
j=json.load(open('gpurun_out/r4p/bench_full.json'))
print('ms', j['ms_per_step'], 'value', j['value'], 'exact', j['exact_fp32_ms_per_step'], j['exact_fp32_rms_diff_of_output'], j.get('exact_fp32_gru_phase_form'))
r=j['roofline']; print('roofline', r['kernel'], r['frac'], r['avg_launch_ms'], 'traffic', r['traffic'], str(r['traffic_source'])[:60])
print('in_loop', {k:v for k,v in r['in_loop'].items() if k not in ('where','note','traffic_source')}, str(r['in_loop'].get('traffic_source'))[:50])
print('step', {k:v for k,v in r['step'].items() if k not in ('macs_per_frame','note')})
h=j['host_io']; print('host_io', {k:(round(v,3) if isinstance(v,float) else v) for k,v in h.items() if k!='how'})
print('stream', j['configs']['streaming_4096']['ungated']['ms_per_call'], j['configs']['streaming_4096']['stage_gating']['ms_per_call'], 'o10', j['configs']['df_apply_o10']['frac'])
print('gru', j['rooflines']['dfx_k_gru_rec_h3'].get('alone'), j['rooflines']['dfx_k_gru_rec_h3'].get('under_load'))
print('enqueue', j['enqueue']['ms_per_step_with_free_enqueue_ahead'], 'cpu', j['cpu_baseline']['value'])
print('kernels', j['kernels'])
PY
