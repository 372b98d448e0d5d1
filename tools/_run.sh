mkdir -p gpurun_out/r04p
O=gpurun_out/r04p
timeout -k 10 300 python3 tools/refill_bound_gpu.py > $O/refill_gpu.txt 2>&1; echo "rc=$?"; grep -v amdgpu.ids $O/refill_gpu.txt
timeout -k 10 300 python3 tools/refill_bound_gpu.py --topology full --frames 32 >> $O/refill_gpu.txt 2>&1; echo "rc=$?"; grep -v amdgpu.ids $O/refill_gpu.txt | tail -5
