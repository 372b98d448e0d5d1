O=gpurun_out/${1:-r06occ}
mkdir -p $O
X=tools/bin/exp
echo "== occupancy experiment: 896 threads x 2 per CU at 7 waves per SIMD (72 VGPRs, spills), bench batch full topology" >> $O/occ.txt
RDF_HIP_LIBRARY=$X/lib_base.so python3 tools/sweep.py --rounds 5 --check 1 0:0 896:81900:2:56:8 896:81900:1:56:9 896:81900:2:48:9 768:81900:2:56:9 >> $O/occ.txt 2>&1
echo "== 1024 threads x 2 per CU at 8 waves per SIMD (64 VGPRs, spills)" >> $O/occ.txt
RDF_HIP_LIBRARY=$X/lib_w8.so python3 tools/sweep.py --rounds 5 --check 1 0:0 1024:81900:2:56:8 1024:81900:1:56:9 1024:81900:2:40:9 >> $O/occ.txt 2>&1
echo "== trained topology" >> $O/occ.txt
RDF_HIP_LIBRARY=$X/lib_base.so python3 tools/sweep.py --rounds 5 --topology trained 0:0 896:81900:2:56:8 >> $O/occ.txt 2>&1
grep -v amdgpu $O/occ.txt
