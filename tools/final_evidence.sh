# the round's final evidence, one gpurun call: GPU suite, the default bench run (full result + compact line), the two kernel traces
O=gpurun_out/${1:-r06final}
mkdir -p $O
python __graft_entry__.py smoke > $O/smoke.log 2>&1; tail -1 $O/smoke.log | cut -c1-200
python -m pytest tests -m gpu -x -q > $O/gputest.log 2>&1; tail -3 $O/gputest.log
python3 bench.py --full-json $O/bench_full.json > $O/bench_line.json 2> $O/bench.err; tail -c 200 $O/bench_line.json; echo
bash tools/profile.sh ${1:-r06final}/prof --deep-from 0 > $O/prof.log 2>&1; tail -2 $O/prof.log | cut -c1-300
DF=$(python3 -c "import json;print(json.load(open('$O/bench_full.json'))['cfg2_balanced']['tune']['deep_from'])")
bash tools/profile.sh ${1:-r06final}/prof_bal --topology balanced --deep-from $DF > $O/prof_bal.log 2>&1; tail -2 $O/prof_bal.log | cut -c1-300
