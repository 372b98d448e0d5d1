#!/usr/bin/env python3
"""Hierarchical roofline of the forest kernel: HBM, L2->L1 line fills, the L1/texture-addresser pipeline and VALU issue.

SURVEY 8(d) prices the kernel against HBM with ALGORITHMIC bytes (32 B per node visit).  Most of those bytes are served
by LDS, L1 and L2, so that figure exceeds the HBM peak (round 1: 1.84x) and bounds nothing.  This module turns the
rocprofv3 counters of one launch into the time each level of the memory/issue hierarchy needs for that launch at its own
measured peak; the level that needs the largest share of the kernel's duration is the bound, and that share is
`roofline.frac` (in (0, 1] by construction, up to the accuracy of the calibration).

Calibration constants (profiles/r02_ubench_*.txt, tools/ubench_valu.hip, tools/ubench_fetch.hip, tools/ubench_vmem.hip;
MI355X_MICROARCH.md for the peaks):
  * FETCH_SIZE = TCC_EA0_RDREQ x 64 B although every request of this kernel's patterns (2-byte probes, 16-byte records,
    coalesced staging) moves a whole 128-byte line: HBM-side read bytes = TCC_EA0_RDREQ x 128 (= 2 x FETCH_SIZE);
    WRITE_SIZE is exact.  Infinity-Cache hits are included, so this is an upper bound of what HBM itself delivers.
  * TCP_TCC_READ_REQ counts 128-byte line requests (one per line for every access width): L2->L1 bytes = requests x 128,
    peak 64 B/clk/CU.
  * L1/TA pipeline: cycles a CU's vector-memory pipeline needs per 128-byte line access of a divergent load, as a
    function of the share of those accesses that are filled from L2 (tools/ubench_l1_fill.hip, shader clock measured
    in the run with s_memtime / s_memrealtime = 2.39 GHz; the share read back with TCP_TCC_READ_REQ /
    TCP_TOTAL_CACHE_ACCESSES): 0.59 cycles when the L1 serves everything, 2.28 when everything is a fill, and in
    between close to max(0.59, 0.08 + 2.24 x share) -- fills and hits overlap, the fill path (2.24 cycles per line =
    57 B/clk/CU of the 64 B/clk/CU L2->L1 interface) is the limiter from a share of a quarter up.  Round 1's two
    constants (0.6 per hit + 2.35 per fill, added up, cycles derived from wall time at an assumed 2.4 GHz) overstated
    this level by 15-25 % at the kernel's fill share of about one half and put it above 1.0 once the kernel got faster;
    the curve agrees with the hardware's own TA_TA_BUSY (0.78 against 0.79 on the bench batch).
  * The memory side's CEILING for the walk's own access pattern -- every lane one random 128-byte line, the next one depending
    on the data (tools/ubench_gather.hip, profiles/r05_ubench_gather.txt): 7.1-7.2 TB/s from a 146-MB and a 1.2-GB table, 6.5 TB/s
    from a 4-GB one, with the wave fetching its lines by LDS-DMA (16 or 8 waves per CU) as with one load per lane (24 waves):
    the `hbm` level carries `frac_of_gather_ceiling` beside its fraction of the 8 TB/s data-sheet peak.
  * TA_TA_BUSY against the L1 level: the counter reads what the L1 level's calibrated cycles say (within 9 %, a little above) as long
    as the fabric is far from its ceiling, and rises steeply as the launch approaches it -- the texture addresser's queues fill
    up behind the misses.  Measured with the gather microbenchmark throttled by arithmetic between two fetches
    (tools/calibrate_ta_busy.sh, profiles/r05_ta_busy_per_miss.txt): busy cycles per L2 miss BEYOND the L1 level's own cycles,
    by the fraction of the gather ceiling the launch runs at -- the wave's cooperative LDS-DMA fetch 2.5 / 2.4 / 2.2 / 1.9 / 6.5
    at 0.13 / 0.24 / 0.42 / 0.68 / 0.98: the curve the deep-block kernels are priced with, nothing fitted.  For the heap-order
    kernels (divergent 16-byte record loads of which a few per cent miss the L2, at 0.06-0.28 of the ceiling) no microbenchmark
    gives the price: the pure gather's every access is a fill, and there the L1 curve over-prices the access itself (its
    residual is negative below 0.8 of the ceiling, 7.7 at 0.99).  Their 5.6 busy cycles per L2 miss are FITTED on the three
    heap-order legs of profiles/r05_bench.json (the L1 level alone reads 3 / 8 / 14 % under the counter on them, in the order
    of their miss counts).  `ta_busy_model` = the L1 level's cycles + TCC_MISS x that price: within 8 % of the counter on all
    five legs -- round 4's model, the L1 level alone, read 0.60 and 0.46 where the counter read 0.94 and 0.97 (the seven-load
    walk ran AT the texture addresser's limit; the cooperative walk runs at 0.71-0.84 of the fabric's).
  * VALU issue with more than one wave per SIMD: 2.35 cycles for the simple integer/fp32 instructions, 4.2 cycles for
    v_pk_*_f32, conversions, v_perm_b32 and the three-operand integer forms (tools/ubench_valu.hip); the kernel's mix is
    counted in its ISA (VALU_MIX below).
"""
import csv
import glob
import json
import os
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

HBM_PEAK_GBS = 8000.0          # spec, MI355X_MICROARCH.md
GATHER_CEILING_GBS = 7150.0    # random dependent 128-byte line gather, tables of 146 MB to 1.2 GB (profiles/r05_ubench_gather.txt)
# profiles/r05_ta_busy_per_miss.txt: (fraction of the gather ceiling, TA busy cycles per L2 miss beyond the L1 level's cycles)
TA_BUSY_CYCLES_PER_L2_MISS = {"deep": [(0.0, 2.5), (0.13, 2.5), (0.24, 2.39), (0.42, 2.18), (0.68, 1.94), (0.98, 6.46), (1.0, 6.46)],
                              "divergent": [(0.0, 5.6), (0.80, 5.6), (0.99, 7.74), (1.0, 7.74)]}


def ta_busy_cycles_per_l2_miss(kind, ceiling_fraction):
    pts = TA_BUSY_CYCLES_PER_L2_MISS[kind]
    u = min(max(ceiling_fraction, 0.0), 1.0)
    for (x0, y0), (x1, y1) in zip(pts, pts[1:]):
        if u <= x1:
            return y0 + (y1 - y0) * (u - x0) / (x1 - x0) if x1 > x0 else y1
    return pts[-1][1]
L2_L1_BYTES_PER_CLK_PER_CU = 64.0
LINE = 128
CUS = 256
SIMDS = 4 * CUS
# profiles/r02_ubench_l1_fill.txt: (share of L1 line accesses filled from L2, cycles per line access per CU); the same
# for 2- and 16-byte loads and for 5 x 256 and 3 x 512 threads per CU
L1_CYCLES_PER_ACCESS = [(0.0, 0.592), (0.171, 0.633), (0.315, 0.827), (0.452, 1.097), (0.573, 1.354), (0.811, 1.881),
                        (0.984, 2.285), (1.0, 2.32)]
VALU_FULL_RATE_CYCLES = 2.35   # profiles/r02_ubench_valu.txt
VALU_HALF_RATE_CYCLES = 4.2
# share of the walk loop's VALU instructions that issue at the half rate (v_pk_*, v_cvt_*, v_perm, v_mad_u32_u24,
# v_lshl_add_u32, v_add_lshl_u32, v_or3, v_bfe, v_mul_u32_u24 ...), counted in the ISA of the headline kernel
VALU_HALF_RATE_SHARE = 0.78     # tools/valu_mix.py on k_eval_forest<512,true,4,false,4,false,1,false>: 150 of 191 (round 2: 215 of 265)

PASSES = [
    ("fetch", "FETCH_SIZE"),
    ("write", "WRITE_SIZE TCC_EA0_RDREQ_sum"),
    ("tcc", "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"),
    ("tcp", "TCP_TCC_READ_REQ_sum"),
    ("tcpacc", "TCP_TOTAL_CACHE_ACCESSES_sum"),
    ("sq", "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"),
    ("grbm", "GRBM_GUI_ACTIVE"),
    ("ta", "TA_TA_BUSY_sum"),
]


def parse_counters(root, kernel_substr, last_n=None):
    """Mean of every counter over the dispatches of the most frequently launched kernel whose name contains
    `kernel_substr` (rocprofv3 --pmc --output-format csv trees under `root`).  With `last_n`: over the LAST n dispatches of
    kernels whose name contains `kernel_substr` in each pass -- the timed launches of a program that first tries other
    launches of the same kernel family (DecisionForest.tune); the kernel named is the one those dispatches ran."""
    acc = {}
    for f in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
        with open(f) as fh:
            rows = [row for row in csv.DictReader(fh) if kernel_substr in row.get("Kernel_Name", "")]
        if last_n:
            ids = sorted({int(row["Dispatch_Id"]) for row in rows})[-last_n:]
            rows = [row for row in rows if int(row["Dispatch_Id"]) in ids]
        for row in rows:
            d = acc.setdefault(row["Kernel_Name"], {})
            d.setdefault(row["Counter_Name"], []).append(float(row["Counter_Value"]))
            try:    # the dispatch's own duration in this (profiled) pass, for the clock estimate
                d.setdefault("_ns:" + row["Counter_Name"], []).append(float(row["End_Timestamp"]) - float(row["Start_Timestamp"]))
                d.setdefault("_vgpr", []).append(float(row["VGPR_Count"]))
                d.setdefault("_lds_bytes", []).append(float(row["LDS_Block_Size"]))
                d.setdefault("_grid", []).append(float(row["Grid_Size"]))
            except (KeyError, ValueError):
                pass
    if not acc:
        return None, {}
    name = max(acc, key=lambda k: max(len(v) for v in acc[k].values()))
    return name, {c: sum(v) / len(v) for c, v in acc[name].items()}


def collect(cmd, kernel_substr, timeout=240, passes=PASSES, keep_dir=None, last_n=None):
    """Runs `cmd` (a list, the program itself first: rocprofv3 must not be handed a launcher) once per counter set under
    rocprofv3 --pmc and returns (kernel name, {counter: mean per launch}, log lines).  Each pass is its own process; a
    pass that fails or times out is skipped.  Call this BEFORE the calling process touches the GPU."""
    if shutil.which("rocprofv3") is None:
        return None, {}, ["rocprofv3 not found"]
    out, log, name = {}, [], None
    base = keep_dir or tempfile.mkdtemp(prefix="rdf_pmc_", dir="/tmp")
    env = dict(os.environ, TMPDIR="/tmp")
    for tag, ctrs in passes:
        d = os.path.join(base, tag)
        full = ["rocprofv3", "--pmc"] + ctrs.split() + ["--output-format", "csv", "-d", d, "--"] + list(cmd)
        try:
            r = subprocess.run(full, cwd="/tmp", env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=timeout)
            log.append(f"pass {tag}: rc={r.returncode}")
            if r.returncode != 0:
                log.append(r.stderr.decode(errors="replace")[-400:])
                continue
        except Exception as e:  # timeout, missing tool
            log.append(f"pass {tag}: {type(e).__name__}: {e}")
            continue
        k, vals = parse_counters(d, kernel_substr, last_n)
        if vals:
            name = name or k
            out.update(vals)
    if keep_dir is None:
        shutil.rmtree(base, ignore_errors=True)
    return name, out, log


def l1_cycles_per_access(share):
    pts = L1_CYCLES_PER_ACCESS
    for (x0, y0), (x1, y1) in zip(pts, pts[1:]):
        if share <= x1:
            return y0 + (y1 - y0) * (share - x0) / (x1 - x0)
    return pts[-1][1]


def is_deep_kernel(kernel_name):
    """k_eval_forest<BLOCK, PACKED, CMAX, STATS, GROUP, COMPACT, NL, TW, DEEP>: the last template argument."""
    if not kernel_name or "k_eval_forest<" not in kernel_name:
        return False
    args = kernel_name.split("k_eval_forest<", 1)[1].split(">", 1)[0].split(",")
    return len(args) >= 9 and args[8].strip() == "true"


def model(counters, kernel_ms, alg_bytes=None, cus=CUS, useful=None, kernel_name=None):
    """The four levels for one launch; returns the `roofline` object of bench.py's JSON line.

    `useful` (rdf_eval_forest_packed_stats, the same launch with counters on): {"records", "leaf_rows", "far_probes",
    "blocks"} = 128-byte lines the loads that serve walking slots touch at the least.  With it the L1 level carries
    `useful_frac`: those lines x the calibrated cycles per access against the kernel's cycles -- a useful-work fraction; the
    level's own `frac` prices the accesses the hardware counted (TCP_TOTAL_CACHE_ACCESSES), i.e. it is a utilisation: a
    kernel that issued twice the accesses it needs would show the same `frac` and half the `useful_frac`."""
    t = kernel_ms * 1e-3
    c = counters or {}
    levels = {}
    clk = None
    if c.get("GRBM_GUI_ACTIVE"):
        # rocprofv3 sums the 8 XCDs; divided by the dispatch's duration in the SAME (profiled) pass when the CSV has it
        t_prof = c.get("_ns:GRBM_GUI_ACTIVE", 0.0) * 1e-9
        clk = c["GRBM_GUI_ACTIVE"] / 8.0 / (t_prof if t_prof > 0 else t)
    clk_used = clk or 2.3e9
    cyc = clk_used * t                                 # kernel duration in shader cycles

    # ---- HBM (fabric side of L2; includes Infinity-Cache hits) ----
    hbm = None
    rd = c.get("TCC_EA0_RDREQ_sum")
    if rd is not None:
        rd_bytes = rd * LINE
    elif c.get("FETCH_SIZE") is not None:
        rd_bytes = c["FETCH_SIZE"] * 1024.0 * 2.0      # 128-byte requests tallied at 64 B
    else:
        rd_bytes = None
    if rd_bytes is not None and c.get("WRITE_SIZE") is not None:
        hbm = rd_bytes + c["WRITE_SIZE"] * 1024.0
        a = hbm / t / 1e9
        levels["hbm"] = {"achieved": round(a, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(a / HBM_PEAK_GBS, 4),
                         "gather_ceiling": GATHER_CEILING_GBS, "frac_of_gather_ceiling": round(a / GATHER_CEILING_GBS, 4),
                         "bytes_per_launch": int(hbm),
                         "what": "L2 fabric-side read requests x 128 B + WRITE_SIZE (Infinity-Cache hits included)"}
    # ---- L2 -> L1 line fills ----
    fills = c.get("TCP_TCC_READ_REQ_sum")
    if fills is not None:
        a = fills * LINE / t / 1e9
        peak = L2_L1_BYTES_PER_CLK_PER_CU * cus * clk_used / 1e9
        levels["l2_l1"] = {"achieved": round(a, 1), "peak": round(peak, 1), "unit": "GB/s", "frac": round(a / peak, 4),
                           "line_fills_per_launch": int(fills),
                           "what": "TCP_TCC_READ_REQ x 128 B against 64 B/clk/CU at the measured clock"}
    # ---- L1 / texture-addresser pipeline ----
    acc = c.get("TCP_TOTAL_CACHE_ACCESSES_sum")
    if fills is not None and acc is not None:
        share = min(fills / acc, 1.0) if acc else 0.0
        ta_cycles = acc * l1_cycles_per_access(share) / cus
        levels["l1_ta"] = {"achieved": round(ta_cycles / 1e6, 3), "peak": round(cyc / 1e6, 3), "unit": "Mcycles per CU",
                           "frac": round(ta_cycles / cyc, 4), "l1_line_accesses_per_launch": int(acc),
                           "fill_share": round(share, 4), "cycles_per_line_access": round(l1_cycles_per_access(share), 3),
                           # cross-check from the hardware's own busy counter (cycles the texture addresser is busy,
                           # summed over the CUs, in the profiled pass that collected it)
                           "ta_busy_frac_counter": (round(c["TA_TA_BUSY_sum"] / cus / (c["_ns:TA_TA_BUSY_sum"] * 1e-9 * clk_used), 4)
                                                    if c.get("TA_TA_BUSY_sum") and c.get("_ns:TA_TA_BUSY_sum") else None),
                           "what": "L1 line accesses x the measured cycles per access at this launch's fill share "
                                   "(tools/ubench_l1_fill.hip) per CU against the kernel's cycles"}
        if c.get("TCC_MISS_sum") is not None and hbm is not None:
            kind = "deep" if is_deep_kernel(kernel_name) else "divergent"
            per_miss = ta_busy_cycles_per_l2_miss(kind, hbm / t / 1e9 / GATHER_CEILING_GBS)
            busy = (ta_cycles + c["TCC_MISS_sum"] * per_miss / cus) / cyc
            levels["l1_ta"].update({"ta_busy_model": round(busy, 4), "ta_busy_cycles_per_l2_miss": round(per_miss, 2), "ta_busy_model_what":
                                    f"the level's cycles + {per_miss:.2f} busy cycles per L2 miss ({kind} loads at this launch's fraction of the "
                                    "gather ceiling: profiles/r05_ta_busy_per_miss.txt): what TA_TA_BUSY should read"})
        if useful:
            lines = float(sum(useful.get(k, 0) for k in ("records", "leaf_rows", "far_probes", "blocks")))
            levels["l1_ta"].update({
                "useful_line_accesses_per_launch": int(lines), "useful_by_kind": {k: int(v) for k, v in useful.items()},
                "useful_share_of_issued": round(lines / acc, 4) if acc else None,
                "useful_frac": round(lines * l1_cycles_per_access(share) / cus / cyc, 4),
                "useful_what": "128-byte lines the loads serving walking slots must touch with this launch geometry (node records "
                               "below the LDS levels, leaf rows, far probes that load, deep blocks; neighbouring lanes on one line "
                               "merged) x the same cycles per access; the rest of the issued accesses is staging, finished slots' "
                               "stand-in fetches and the second to seventh load of a line"})
    # ---- VALU issue ----
    valu = c.get("SQ_INSTS_VALU")
    if valu is not None:
        per = VALU_HALF_RATE_SHARE * VALU_HALF_RATE_CYCLES + (1 - VALU_HALF_RATE_SHARE) * VALU_FULL_RATE_CYCLES
        v_cycles = valu * per / (4 * cus)
        levels["valu"] = {"achieved": round(v_cycles / 1e6, 3), "peak": round(cyc / 1e6, 3), "unit": "Mcycles per SIMD",
                          "frac": round(v_cycles / cyc, 4), "valu_wave_instructions_per_launch": int(valu),
                          "what": f"SQ_INSTS_VALU x {per:.2f} cycles ({VALU_HALF_RATE_SHARE:.0%} half-rate instructions, "
                                  "tools/ubench_valu.hip) per SIMD against the kernel's cycles"}
    out = {"kernel_ms": round(kernel_ms, 4), "clock_ghz": round(clk / 1e9, 3) if clk else None, "traffic": int(hbm) if hbm else None,
           "levels": levels}
    if c.get("TCC_REQ_sum"):
        out["l2_hit_rate"] = round(c.get("TCC_HIT_sum", 0.0) / c["TCC_REQ_sum"], 4)
    if levels:
        b = max(levels, key=lambda k: levels[k]["frac"])
        out.update({"bound": b, "achieved": levels[b]["achieved"], "peak": levels[b]["peak"], "unit": levels[b]["unit"],
                    "frac": levels[b]["frac"]})
        if "useful_frac" in levels.get("l1_ta", {}):
            out["useful_frac"] = levels["l1_ta"]["useful_frac"] if b == "l1_ta" else levels[b]["frac"]
            out["useful_frac_what"] = ("the bounding level's fraction counted in useful work: for the L1 / texture-addresser level the "
                                       "lines the walk needs instead of the accesses the kernel issued; the HBM, L2->L1 and VALU levels "
                                       "count what the hardware moved or issued")
    else:
        out.update({"bound": None, "achieved": None, "peak": None, "unit": None, "frac": None})
    if alg_bytes is not None:
        a = alg_bytes / t / 1e9
        out["algorithmic"] = {"bytes_per_launch": int(alg_bytes), "rate_gbs": round(a, 1),
                              "over_hbm_peak": round(a / HBM_PEAK_GBS, 4),
                              "what": "SURVEY 8(d): 32 B per node visit + 4C per leaf + 4 B/px; mostly served by LDS/L1/L2, "
                                      "so this rate is not bounded by HBM"}
    return out


if __name__ == "__main__":
    # usage: tools/roofline.py <dir with rocprofv3 csv trees> <kernel substring> <kernel_ms>
    name, vals = parse_counters(sys.argv[1], sys.argv[2])
    print(json.dumps({"kernel": name, "counters": vals, "roofline": model(vals, float(sys.argv[3]))}, indent=1))
