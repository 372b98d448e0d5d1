#!/usr/bin/env python3
"""Does every rounding fp32 instruction of the forest kernels run in the rounding mode it was written for?

The level loop computes divide + floor + add as ONE fma in round-toward-minus-infinity mode (rdf_hip.hip, NodeRec16); the
reciprocal's refinement, the IEEE divides of flagged nodes and the sums of leaf PDFs need round-to-nearest.  The mode is
switched with s_setreg inside `asm volatile` blocks that are tied to the arithmetic around them by data dependences only
(rdf_device.hpp: set_round_down / set_round_nearest / pin) -- the compiler assumes the default fp environment and may move
an independent fp instruction across a switch.  This tool checks the RESULT: it compiles rdf_hip.hip to gfx950 ISA, builds
the control-flow graph of every k_eval_forest instantiation, propagates the mode (entry: nearest; s_setreg_imm32_b32
hwreg(HW_REG_MODE, 0, 2), 2 / 0) through it and reports
  * an add / sub / mul / fmac / rcp / divide-sequence instruction that can execute in round-down mode,
  * a packed fma (the kernels' packed fmas are the one-fma divide-and-floor) that can execute in round-to-nearest,
  * a scalar fma reached in round-down mode whose block has no 0x4b400000 (1.5 * 2^23, the magic addend) in sight,
  * a block reached in both modes that holds any of them.
v_cvt_f32_i32 may run in either mode (NodeRec16: the guard bits make the decoded numerator's rounding irrelevant).

    python3 tools/check_rounding_isa.py            # exit status 1 and a report when something is out of place
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "3d-beats_amd", "csrc", "rdf_hip.hip")
HDRS = [os.path.join(ROOT, "3d-beats_amd", "csrc", "rdf_device.hpp"), os.path.join(ROOT, "include", "rdf_hip.h")]
NEAREST_ONLY = ("v_add_f32", "v_sub_f32", "v_subrev_f32", "v_mul_f32", "v_pk_add_f32", "v_pk_mul_f32", "v_fmac_f32", "v_rcp_f32",
                "v_div_scale_f32", "v_div_fmas_f32", "v_div_fixup_f32", "v_mac_f32", "v_madak_f32", "v_madmk_f32", "v_fmaak_f32", "v_fmamk_f32")
DOWN_ONLY = ("v_pk_fma_f32",)
MAGIC = "0x4b400000"
EXACT_SCALINGS = ("0x44000000", "0x3b000000")      # x 512 and x 1/512 (kNumScale): exact whatever the mode


def isa_text():
    """ISA of rdf_hip.hip (cached under /tmp by the sources' modification times)."""
    stamp = "_".join(str(int(os.path.getmtime(p))) for p in [SRC] + HDRS)
    out = os.path.join(tempfile.gettempdir(), f"rdf_hip_gfx950_{stamp}.s")
    if not os.path.exists(out):
        d = tempfile.mkdtemp(prefix="rdf_isa_")
        subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "--offload-arch=gfx950", "-std=c++17", "-fno-fast-math", "-ffp-contract=off",
                               "--cuda-device-only", "-S", "-o", out + ".tmp", SRC], cwd=d, stderr=subprocess.DEVNULL)
        os.replace(out + ".tmp", out)
    return open(out).read()


def kernels(text):
    """(mangled name, body lines) of every k_eval_forest kernel."""
    for m in re.finditer(r"^(_ZN\S*k_eval_forest\S*):.*\n", text, re.M):
        end = text.find(".Lfunc_end", m.end())
        yield m.group(1), text[m.end():end].split("\n")


def analyse(lines):
    """Forward data-flow of the rounding mode over the kernel's basic blocks.  Returns a list of findings."""
    blocks, cur, label_of = [], {"label": "entry", "ins": []}, {}
    for ln in lines:
        t = ln.split(";")[0].strip()
        if not t or t.startswith("."):
            if t.startswith(".LBB") and t.endswith(":"):
                pass
            else:
                continue
        if t.endswith(":"):
            if cur["ins"] or cur["label"] == "entry":
                blocks.append(cur)
            cur = {"label": t[:-1], "ins": []}
            continue
        cur["ins"].append(t)
        op = t.split()[0]
        if op.startswith("s_cbranch") or op == "s_branch" or op == "s_endpgm" or op.startswith("s_setpc"):
            blocks.append(cur)
            cur = {"label": None, "ins": []}
    if cur["ins"]:
        blocks.append(cur)
    for i, b in enumerate(blocks):
        if b["label"]:
            label_of[b["label"]] = i
    succ = []
    for i, b in enumerate(blocks):
        s = []
        last = b["ins"][-1] if b["ins"] else ""
        op = last.split()[0] if last else ""
        tgt = last.split()[-1] if last else ""
        if op == "s_branch":
            s = [label_of[tgt]] if tgt in label_of else []
        elif op.startswith("s_cbranch"):
            s = ([label_of[tgt]] if tgt in label_of else []) + ([i + 1] if i + 1 < len(blocks) else [])
        elif op == "s_endpgm":
            s = []
        else:
            s = [i + 1] if i + 1 < len(blocks) else []
        succ.append(s)

    def transfer(state, b):
        for t in b["ins"]:
            if t.startswith("s_setreg_imm32_b32") and "HW_REG_MODE, 0, 2)" in t:
                state = "down" if t.rstrip().endswith(", 2") else "nearest"
            elif t.startswith("s_setreg") and "HW_REG_MODE" in t:
                state = "unknown"
        return state
    in_state = [None] * len(blocks)
    in_state[0] = "nearest"
    work = [0]
    while work:
        i = work.pop()
        out = transfer(in_state[i], blocks[i])
        for j in succ[i]:
            new = out if in_state[j] in (None, out) else "both"
            if new != in_state[j]:
                in_state[j] = new
                work.append(j)
    findings = []
    for i, b in enumerate(blocks):
        state = in_state[i]
        if state is None:
            continue                                            # unreachable
        has_magic = any(MAGIC in t for t in b["ins"])
        for t in b["ins"]:
            op = t.split()[0]
            if t.startswith("s_setreg_imm32_b32") and "HW_REG_MODE, 0, 2)" in t:
                state = "down" if t.rstrip().endswith(", 2") else "nearest"
                continue
            base = op.replace("_e32", "").replace("_e64", "").replace("_dpp", "").replace("_sdwa", "")
            if base == "v_mul_f32" and any(lit in t for lit in EXACT_SCALINGS):
                continue                                        # a multiplication by a power of two: exact in every mode
            sure = state != "both"
            if base in NEAREST_ONLY and state != "nearest":
                findings.append((sure, f"{base} {'runs' if sure else 'may run'} in mode '{state}' (block {b['label'] or i}): {t}"))
            elif base in DOWN_ONLY and state != "down":
                findings.append((sure, f"{base} {'runs' if sure else 'may run'} in mode '{state}' (block {b['label'] or i}): {t}"))
            elif base == "v_fma_f32" and state == "both":
                findings.append((False, f"v_fma_f32 in a block reached in both modes (block {b['label'] or i}): {t}"))
            elif base == "v_fma_f32" and state == "down" and not has_magic and MAGIC not in t:
                findings.append((True, f"v_fma_f32 in round-down mode without the magic addend in its block (block {b['label'] or i}): {t}"))
    return findings, sum(1 for s in in_state if s == "down"), len(blocks)


def report():
    """[(kernel, [definite findings], [possible findings])] for every k_eval_forest instantiation.

    The analysis is path-insensitive.  Where hipcc threads a jump through a flag (the "level D-1 from the table" test of the
    packed kernels) or a kernel switches the mode depending on the level it is on (the reference-layout kernels: round-down
    for the levels held in LDS, nearest below), a block is reached "in both modes" on paper although every real path enters it in
    one: those are `possible` findings, listed for the reader; `definite` ones -- every path agrees on the wrong mode -- are
    what tests/test_isa_rounding.py refuses."""
    out = []
    for name, lines in kernels(isa_text()):
        findings, n_down, n_blocks = analyse(lines)
        dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
        short = dem[dem.find("k_eval_forest"):dem.find(">(") + 1]
        out.append((short, [f for sure, f in findings if sure], [f for sure, f in findings if not sure], n_down, n_blocks))
    return out


def main():
    bad = 0
    rows = report()
    for short, definite, possible, n_down, n_blocks in rows:
        if definite or possible:
            print(f"{short}: {len(definite)} definite, {len(possible)} possible ({n_down} of {n_blocks} blocks entered in round-down mode)")
        for f in definite:
            print("    DEFINITE " + f)
        if "-v" in sys.argv:
            for f in possible:
                print("    possible " + f)
        bad += 1 if definite else 0
    print(f"{len(rows)} kernels checked, {bad} with definite findings")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
