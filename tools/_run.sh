mkdir -p gpurun_out/r04j
O=gpurun_out/r04j
python3 -m pytest tests -x -q -m gpu -k "deep or tune or balanced" > $O/pytest_deep.txt 2>&1; echo "rc=$?" >> $O/pytest_deep.txt; tail -n 15 $O/pytest_deep.txt
python3 -m pytest tests -x -q -m gpu > $O/pytest_gpu.txt 2>&1; echo "rc=$?" >> $O/pytest_gpu.txt; tail -n 15 $O/pytest_gpu.txt
