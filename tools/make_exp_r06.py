#!/usr/bin/env python3
"""Round 6's timing experiments on the headline kernel, as edits applied to a COPY of csrc/ under tools/bin/exp/ (git-ignored;
it travels to the GPU box with gpurun).  The product source never carries experiment code: this script is the record of what
each variant changed.  Variants (select with -D at build time):

  (always)            workgroups of 768 and 1024 threads accepted by rdf_set_block_threads for packed, unfiltered, four-class,
                      four-trees-in-a-lane launches (labels stay correct: a real candidate geometry)
  RDF_ABL_PDF         last-level pass without the two PDF loads per tree (timing only: labels wrong)
  RDF_ABL_LDSNODES=n  node records of the levels below n come from LDS whatever the LDS table holds (index masked: timing only)
  RDF_ABL_STAGE       tiles are not staged (timing only)
  RDF_ABL_FAR         far probes do not load (timing only)

usage: python3 tools/make_exp_r06.py        # copies, edits, builds tools/bin/exp/lib_<variant>.so
"""
import os
import re
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXP = os.path.join(ROOT, "tools", "bin", "exp")
SRC = os.path.join(EXP, "pkg", "csrc")
FLAGS = ["-O3", "--offload-arch=gfx950", "-std=c++17", "-shared", "-fPIC", "-fno-fast-math", "-ffp-contract=off"]


def sub1(text, old, new):
    assert text.count(old) == 1, (text.count(old), old[:80])
    return text.replace(old, new)


def edit():
    os.makedirs(SRC, exist_ok=True)
    os.makedirs(os.path.join(EXP, "include"), exist_ok=True)
    for f in os.listdir(os.path.join(ROOT, "3d-beats_amd", "csrc")):
        if f.endswith((".hip", ".hpp")):
            shutil.copy(os.path.join(ROOT, "3d-beats_amd", "csrc", f), SRC)
    shutil.copy(os.path.join(ROOT, "include", "rdf_hip.h"), os.path.join(EXP, "include"))
    p = os.path.join(SRC, "rdf_hip.hip")
    t = open(p).read()
    # -- bigger workgroups
    t = sub1(t, "__launch_bounds__(BLOCK, TW ? 8 : DEEP ? 4 : BLOCK == 512 ? 6 : BLOCK == 256 ? 5 : 4)",
             "__launch_bounds__(BLOCK, TW ? 8 : DEEP ? 4 : BLOCK == 512 ? 6 : BLOCK == 256 ? 5 : BLOCK == 768 ? 6 : BLOCK == 896 ? 7 : BLOCK == 1024 ? RDF_EXP_WAVES_1024 : 4)")
    t = sub1(t, "namespace {\n\nconstexpr int kGroup = 4;", "#ifndef RDF_EXP_WAVES_1024\n#define RDF_EXP_WAVES_1024 4\n#endif\nnamespace {\n\nconstexpr int kGroup = 4;")
    t = sub1(t, "if (block != 256 && block != 512) block = (big && !filtered_r1) ? 512 : 256;",
             "if (block != 256 && block != 512 && block != 768 && block != 896 && block != 1024) block = (big && !filtered_r1) ? 512 : 256;\n"
             "    if (block > 512 && (!packed || filter_class != -1 || n_classes > 4 || stats)) block = 512;")
    t = sub1(t, "        rc = block == 512 ? launch_block<512>(packed != nullptr, compact_launch, a, lds_bytes, cus, st)\n",
             "        rc = block == 768 ? launch_one<768, true, 4, false, 4, false>(a, lds_bytes, cus, st)\n"
             "           : block == 896 ? launch_one<896, true, 4, false, 4, false>(a, lds_bytes, cus, st)\n"
             "           : block == 1024 ? launch_one<1024, true, 4, false, 4, false>(a, lds_bytes, cus, st)\n"
             "           : block == 512 ? launch_block<512>(packed != nullptr, compact_launch, a, lds_bytes, cus, st)\n")
    # (rows per wave for the big blocks: the knob; default 2)
    t = sub1(t, "        if (big && block == 512) {", "        if (big && block >= 512) {")
    # -- ablations
    t = sub1(t, "                                float4 pa = *reinterpret_cast<const float4 *>(pp);\n"
                "                                float4 pb = *reinterpret_cast<const float4 *>(pp + 1);\n",
             "#ifdef RDF_ABL_PDF\n"
             "                                float4 pa = make_float4(1.f, 0.f, 0.f, 0.f), pb = make_float4(0.f, 1.f, 0.f, 0.f); (void)pp;\n"
             "#else\n"
             "                                float4 pa = *reinterpret_cast<const float4 *>(pp);\n"
             "                                float4 pb = *reinterpret_cast<const float4 *>(pp + 1);\n"
             "#endif\n")
    t = sub1(t, "                        const bool in_lds = j < K;\n",
             "#ifdef RDF_ABL_LDSNODES\n"
             "                        const bool in_lds = j < RDF_ABL_LDSNODES;\n"
             "#else\n"
             "                        const bool in_lds = j < K;\n"
             "#endif\n")
    t = sub1(t, "                                n[k] = decode_node(lds_nodes[tk * lds_pitch + hn[k]]);\n",
             "#ifdef RDF_ABL_LDSNODES\n"
             "                                n[k] = decode_node(lds_nodes[tk * lds_pitch + (hn[k] & (lds_pitch - 1u))]);\n"
             "#else\n"
             "                                n[k] = decode_node(lds_nodes[tk * lds_pitch + hn[k]]);\n"
             "#endif\n")
    t = sub1(t, "        if (stage_tw8 > 0u) {\n", "#ifdef RDF_ABL_STAGE\n        if (false) {\n#else\n        if (stage_tw8 > 0u) {\n#endif\n")
    open(p, "w").write(t)
    d = os.path.join(SRC, "rdf_device.hpp")
    t = open(d).read()
    t = sub1(t, "        if (x2 < c.W2 && y < c.H)   // only far lanes touch global memory; the value is consumed after the branch\n"
                "            p.glb_v = *reinterpret_cast<const uint16_t *>(c.img_b + (__umul24(y, c.W2) + x2));\n",
             "#ifndef RDF_ABL_FAR\n"
             "        if (x2 < c.W2 && y < c.H)   // only far lanes touch global memory; the value is consumed after the branch\n"
             "            p.glb_v = *reinterpret_cast<const uint16_t *>(c.img_b + (__umul24(y, c.W2) + x2));\n"
             "#else\n"
             "        if (x2 < c.W2 && y < c.H) p.glb_v = 4000u;\n"
             "#endif\n")
    open(d, "w").write(t)


VARIANTS = {"base": [], "w8": ["-DRDF_EXP_WAVES_1024=8"], "pdf": ["-DRDF_ABL_PDF"], "lds9": ["-DRDF_ABL_LDSNODES=9"], "lds11": ["-DRDF_ABL_LDSNODES=11"],
            "stage": ["-DRDF_ABL_STAGE"], "far": ["-DRDF_ABL_FAR"]}


def build(names):
    srcs = [os.path.join(SRC, f) for f in ("rdf_hip.hip", "mean_shift_hip.hip", "points_ops_hip.hip", "tree_train_hip.hip")]
    procs = []
    for n in names:
        out = os.path.join(EXP, f"lib_{n}.so")
        cmd = ["hipcc"] + FLAGS + VARIANTS[n] + ['-DRDF_BUILD_ID="exp-' + n + '"', "-o", out] + srcs
        procs.append((n, subprocess.Popen(cmd)))
        if len(procs) >= 3:
            for m, pr in procs:
                assert pr.wait() == 0, m
            procs = []
    for m, pr in procs:
        assert pr.wait() == 0, m


if __name__ == "__main__":
    edit()
    build(sys.argv[1:] or list(VARIANTS))
