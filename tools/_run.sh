mkdir -p gpurun_out/r04k
O=gpurun_out/r04k
python3 -m pytest tests -x -q -m gpu -k "rccl_path or host_mapped_get or callconv or deep or tune" > $O/pytest_sel.txt 2>&1; echo "rc=$?" >> $O/pytest_sel.txt; tail -n 12 $O/pytest_sel.txt
