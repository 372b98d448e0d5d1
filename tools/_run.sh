mkdir -p gpurun_out/r04r
O=gpurun_out/r04r
python3 -m pytest tests -x -q -m gpu > $O/pytest_gpu.txt 2>&1; echo "rc=$?" >> $O/pytest_gpu.txt; tail -n 6 $O/pytest_gpu.txt
python3 __graft_entry__.py smoke 2>&1 | tail -n 2
