#!/usr/bin/env python3
"""summary.txt (tools/pmc_summary.py) -> roofline_traffic.json entry for bench.py's `roofline.traffic`.

HBM-side bytes per launch = (FETCH_SIZE + WRITE_SIZE) * 1024 (both counters are in KiB).  Per
MI355X_MICROARCH.md the gfx950 FETCH_SIZE under-reports wide coalesced streams by 2x; this kernel's
reads are narrow gathers (16-byte node records, 2-byte probes), a pattern the guide calls
uncalibrated, so the raw sum is reported together with TCC_MISS*64 B as a cross-check."""
import json
import sys


def main(path, key, kernel):
    vals, take = {}, False
    for line in open(path):
        if line.startswith("=="):
            take = kernel in line
            continue
        if take and "mean" in line:
            name, rest = line.split("mean")
            vals[name.strip()] = float(rest.split()[0])
    fetch, write = vals.get("FETCH_SIZE"), vals.get("WRITE_SIZE")
    out = {key: {
        "kernel": kernel,
        "hbm_bytes_per_launch": int((fetch + write) * 1024) if fetch is not None and write is not None else None,
        "fetch_size_kib": fetch, "write_size_kib": write,
        "tcc_miss_x64B": int(vals["TCC_MISS_sum"] * 64) if "TCC_MISS_sum" in vals else None,
        "tcc_req": vals.get("TCC_REQ_sum"), "tcc_hit": vals.get("TCC_HIT_sum"),
        "tcp_cache_accesses": vals.get("TCP_TOTAL_CACHE_ACCESSES_sum"),
        "grbm_gui_active_sum_over_xcds": vals.get("GRBM_GUI_ACTIVE"),
        "sq_insts_valu": vals.get("SQ_INSTS_VALU"), "sq_insts_salu": vals.get("SQ_INSTS_SALU"),
        "sq_insts_vmem_rd": vals.get("SQ_INSTS_VMEM_RD"), "sq_insts_lds": vals.get("SQ_INSTS_LDS"),
        "sq_wave_cycles": vals.get("SQ_WAVE_CYCLES"), "sq_wait_any": vals.get("SQ_WAIT_ANY"),
        "sq_wait_inst_any": vals.get("SQ_WAIT_INST_ANY"), "sq_active_inst_any": vals.get("SQ_ACTIVE_INST_ANY"),
        "note": "FETCH_SIZE uncalibrated for narrow gathers on gfx950 (see MI355X_MICROARCH.md, HBM)"}}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main(*sys.argv[1:4])
