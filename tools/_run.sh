mkdir -p gpurun_out/r04n
O=gpurun_out/r04n
bash tools/profile.sh r04n/prof --deep-from 0 > $O/profile.log 2>&1; head -n 4 gpurun_out/r04n/prof/kernel_stats.csv; tail -c 300 gpurun_out/r04n/prof/bench_traced.json | head -c 300; echo
bash tools/profile.sh r04n/prof_bal --topology balanced --deep-from 15 > $O/profile_bal.log 2>&1; head -n 4 gpurun_out/r04n/prof_bal/kernel_stats.csv
( RDF_FUZZ_ROUNDS=1500 RDF_LAYERED_FUZZ_ROUNDS=300 RDF_LAST_LEVEL_FUZZ_ROUNDS=300 RDF_FUZZ_SEED=404 timeout -k 10 700 python3 -m pytest tests/test_gpu_parity.py -q -m gpu -k fuzz > $O/fuzz_soak.log 2>&1; echo "rc=$?" >> $O/fuzz_soak.log ) ; tail -n 4 $O/fuzz_soak.log
