mkdir -p gpurun_out/r04m
O=gpurun_out/r04m
python3 -m pytest tests -x -q -m gpu > $O/pytest_gpu.txt 2>&1; echo "rc=$?" >> $O/pytest_gpu.txt; tail -n 6 $O/pytest_gpu.txt
python3 bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; tail -c 400 $O/bench.err
bash tools/profile.sh r04m/prof > $O/profile.log 2>&1; tail -n 8 $O/profile.log
bash tools/profile.sh r04m/prof_bal --topology balanced > $O/profile_bal.log 2>&1; tail -n 8 $O/profile_bal.log
python3 - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/r04m/bench.json') if l.startswith('{')][-1])
print('value', d['value'], d['ms_per_step'], 'bound', d.get('bound'), 'hbm', d.get('hbm_frac'), 'valid', d.get('value_valid_pixels'), 'balanced', d.get('value_balanced'))
r=d['roofline']; print('headline useful', r.get('useful_frac'), r['levels']['l1_ta'].get('useful_share_of_issued'), r['levels']['l1_ta'].get('useful_by_kind'))
for k in ('cfg2_balanced','cfg5_shard','cfg5_balanced'):
    v=d.get(k,{}); r=(v.get('batch') or v).get('roofline')
    print(k, v.get('error'), (v.get('batch') or v).get('value'), r and (r.get('bound'), r.get('frac'), r.get('useful_frac'), r['levels']['l1_ta'].get('useful_share_of_issued'), r['levels']['l1_ta'].get('ta_busy_frac_counter')))
PY
