mkdir -p gpurun_out/r04o
O=gpurun_out/r04o
timeout -k 10 500 python3 tools/trained_forest_probe.py > $O/trained_probe.txt 2>&1; echo "rc=$?"; grep -v amdgpu.ids $O/trained_probe.txt | cut -c1-600
( RDF_FUZZ_ROUNDS=4000 RDF_LAYERED_FUZZ_ROUNDS=800 RDF_LAST_LEVEL_FUZZ_ROUNDS=800 RDF_FUZZ_SEED=20261005 timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py -q -m gpu -k fuzz > $O/fuzz_soak2.log 2>&1; echo "rc=$?" >> $O/fuzz_soak2.log ) ; tail -n 3 $O/fuzz_soak2.log
