#!/bin/bash
# Round 6's experiments on the headline kernel (run through gpurun after `python3 tools/make_exp_r06.py`, which builds the
# variants from a COPY of the sources under tools/bin/exp/): candidate geometries with bigger workgroups (labels checked), the
# timing-only ablations, and the occupancy experiment (7 / 8 waves per SIMD with the spills they cost).
# -> gpurun_out/<tag>/sweep.txt      (profiles/r06_headline_ablation.txt is the record of the round's runs)
O=gpurun_out/${1:-r06exp}
mkdir -p $O
X=tools/bin/exp
echo "== base library (768/896/1024-thread workgroups allowed), bench batch, full topology" >> $O/sweep.txt
RDF_HIP_LIBRARY=$X/lib_base.so python3 tools/sweep.py --rounds 5 --check 2 0:0 768:81900:2:56:9 768:81900:2:64:8 768:81900:2:48:9 768:81900:2:72:7 768:81900:4:56:8 1024:163000:2:56:10 1024:163000:2:72:10 512:81900:2:56:9 512:54600:2:48:8 >> $O/sweep.txt 2>&1
echo "== the same library, trained-like topology" >> $O/sweep.txt
RDF_HIP_LIBRARY=$X/lib_base.so python3 tools/sweep.py --rounds 5 --topology trained 0:0 768:81900:2:56:9 768:81900:2:72:7 >> $O/sweep.txt 2>&1
echo "== the same library, config 5's shard shape (32 dense 1280x720, T8/D22 full)" >> $O/sweep.txt
RDF_HIP_LIBRARY=$X/lib_base.so python3 tools/sweep.py --rounds 4 --trees 8 --depth 22 --frames 32 --height 720 --width 1280 --kinds dense --check 1 0:0 768:81900:2:56:8 768:81900:2:72:6 >> $O/sweep.txt 2>&1
for V in pdf lds9 lds11 stage far; do
echo "== ablation $V (timing only), default geometry" >> $O/sweep.txt
RDF_HIP_LIBRARY=$X/lib_$V.so python3 tools/sweep.py --rounds 5 --no-compare 0:0 >> $O/sweep.txt 2>&1
done
echo "== occupancy: 896 threads x 2 per CU at 7 waves per SIMD (72 VGPRs: 33 spilled dwords)" >> $O/sweep.txt
RDF_HIP_LIBRARY=$X/lib_base.so python3 tools/sweep.py --rounds 5 --check 1 0:0 896:81900:2:56:8 896:81900:1:56:9 >> $O/sweep.txt 2>&1
echo "== occupancy: 1024 threads x 2 per CU at 8 waves per SIMD (64 VGPRs: 62 spilled dwords)" >> $O/sweep.txt
RDF_HIP_LIBRARY=$X/lib_w8.so python3 tools/sweep.py --rounds 5 --check 1 0:0 1024:81900:2:56:8 1024:81900:1:56:9 >> $O/sweep.txt 2>&1
grep -v amdgpu.ids $O/sweep.txt
