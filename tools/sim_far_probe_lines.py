#!/usr/bin/env python3
"""How many 128-byte lines does one wave-level far-probe instruction of the forest kernel touch, with the frame stored
row-major (as the caller hands it over) and with an 8x8-tiled copy (one line = an 8x8 patch), for a wave that covers 64
consecutive pixels of a row and for one that covers an 8x8 block?

A simulation on the bench workload itself (dense synthetic 848x480 frame, T4/D20 "full" forest): the walk of sampled
waves is replayed in numpy (fp32 offsets as decision_tree_common.hpp:15-22), every probe that leaves the staged LDS
tile (64 x 16 pixels + halo) is a far probe, and the distinct lines per (level, tree, probe) instruction are counted.
With an L1 that turns over every few hundred cycles this count is what TCP_TCC_READ_REQ sees.

    python3 tools/sim_far_probe_lines.py [--waves 300] [--halo 32]
"""
import argparse
import importlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def lines_row_major(x, y, w):
    return (y.astype(np.int64) * w + x) * 2 // 128


def lines_tiled(x, y, w):
    tiles_x = (w + 7) // 8
    return (y.astype(np.int64) // 8) * tiles_x + x // 8


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--waves", type=int, default=300)
    ap.add_argument("--halo", type=int, default=32)
    ap.add_argument("--trees", type=int, default=4)
    ap.add_argument("--depth", type=int, default=20)
    a = ap.parse_args()
    synth = importlib.import_module("3d-beats_amd.synth")
    H, W = 480, 848
    frame = synth.dense_frame(0, H, W)
    forest = synth.forest(a.trees, a.depth, 4, "full")
    rng = np.random.default_rng(5)
    out = {}
    for shape in ("row of 64", "8x8 block"):
        tot = {"row-major": 0, "8x8-tiled": 0}
        far_lanes = instr = 0
        for _ in range(a.waves):
            if shape == "row of 64":
                x0, y0 = int(rng.integers(0, W - 64)), int(rng.integers(0, H))
                px, py = x0 + np.arange(64), np.full(64, y0)
                # the workgroup's staged tile: 64 columns x 16 rows around the wave's row, + halo
                tx0, ty0, tw, th = x0 - a.halo, (y0 // 16) * 16 - a.halo, 64 + 2 * a.halo, 16 + 2 * a.halo
            else:
                x0, y0 = int(rng.integers(0, W - 8)), int(rng.integers(0, H - 8))
                gy, gx = np.mgrid[0:8, 0:8]
                px, py = (x0 + gx).reshape(-1), (y0 + gy).reshape(-1)
                tx0, ty0, tw, th = (x0 // 32) * 32 - a.halo, (y0 // 32) * 32 - a.halo, 32 + 2 * a.halo, 32 + 2 * a.halo
            d = frame[py, px].astype(np.float32)
            for t in range(a.trees):
                g = np.zeros(64, np.int64)
                for j in range(a.depth):
                    node = forest[t, (1 << j) - 1 + g]
                    for k in (0, 2):
                        ox = np.floor(node[:, k] / d).astype(np.int64)
                        oy = np.floor(node[:, k + 1] / d).astype(np.int64)
                        qx, qy = px + ox, py + oy
                        in_tile = (qx >= tx0) & (qx < tx0 + tw) & (qy >= ty0) & (qy < ty0 + th)
                        in_img = (qx >= 0) & (qx < W) & (qy >= 0) & (qy < H)
                        far = ~in_tile & in_img
                        if far.any():
                            instr += 1
                            far_lanes += int(far.sum())
                            tot["row-major"] += len(np.unique(lines_row_major(qx[far], qy[far], W)))
                            tot["8x8-tiled"] += len(np.unique(lines_tiled(qx[far], qy[far], W)))
                    ux = np.clip(px + np.floor(node[:, 0] / d).astype(np.int64), -1, W)
                    uy = np.clip(py + np.floor(node[:, 1] / d).astype(np.int64), -1, H)
                    vx = np.clip(px + np.floor(node[:, 2] / d).astype(np.int64), -1, W)
                    vy = np.clip(py + np.floor(node[:, 3] / d).astype(np.int64), -1, H)

                    def val(xx, yy):
                        ok = (xx >= 0) & (xx < W) & (yy >= 0) & (yy < H)
                        v = np.full(64, 65535.0, np.float32)
                        v[ok] = frame[yy[ok], xx[ok]]
                        return v
                    f = val(ux, uy) - val(vx, vy)
                    g = 2 * g + (f >= node[:, 4]).astype(np.int64)
        out[shape] = {"far_probe_instructions": instr, "far_lanes": far_lanes,
                      "far_lanes_per_instruction": round(far_lanes / max(1, instr), 2),
                      "lines_row_major": tot["row-major"], "lines_8x8_tiled": tot["8x8-tiled"],
                      "tiled_over_row_major": round(tot["8x8-tiled"] / max(1, tot["row-major"]), 3)}
    base = out["row of 64"]["lines_row_major"] / out["row of 64"]["far_lanes"]
    for shape in out:
        out[shape]["lines_per_far_lane_row_major"] = round(out[shape]["lines_row_major"] / out[shape]["far_lanes"], 3)
        out[shape]["lines_per_far_lane_tiled"] = round(out[shape]["lines_8x8_tiled"] / out[shape]["far_lanes"], 3)
    out["note"] = (f"{a.waves} sampled waves per shape, dense frame #0, T{a.trees}/D{a.depth} full forest, halo {a.halo}; "
                   f"today's kernel = 'row of 64' on the row-major frame: {base:.3f} distinct lines per far lane")
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
