"""Generates tests/golden/rdf_golden_v1.npz with the CPU oracle (oracle/rdf_oracle.c).

The reference cannot be run (CUDA only) and ships no fixtures, so these vectors are produced by
this repo's restatement of its kernels; they pin the oracle against regressions and are the
committed expected outputs the HIP path is checked against on the GPU box.  Inputs are stored in
full (not just seeds) so the vectors stay valid if the synthetic generators ever change.

Run from the repository root:  python tests/golden/make_golden.py
"""
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import rdf_oracle as oc  # noqa: E402

synth = importlib.import_module("3d-beats_amd.synth")
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "rdf_golden_v1.npz")
COND = np.array([(0, 1), (0, 2), (1, 3), (0, 3), (0, 4)], dtype=np.int32)  # decision_tree.py:214-220


def main():
    z = {}
    # g1: flat forest, full topology (dyadic PDFs: valid against any summation order), batch of 2
    d = synth.frames(["dense", "live"], 100, 48, 64)
    f = synth.forest(4, 8, 4, "full")
    lab = np.full((2, 48, 64), 65535, np.uint16)
    oc.eval_forest(d, f, lab)
    z.update(g1_depth=d, g1_forest=f, g1_labels=lab)

    # g2: trained-like topology, labels_reduce 2, scale 0.5, filter image, pre-fill 0
    d = synth.frames(["live", "dense", "live"], 200, 50, 70)
    f = synth.forest(3, 9, 5, "trained", first_tree=10)
    filt = (np.random.default_rng(5).integers(0, 3, size=(3, 25, 35))).astype(np.uint16)
    lab = np.zeros((3, 25, 35), np.uint16)
    oc.eval_forest(d, f, lab, 2, filt, 1, 0.5)
    z.update(g2_depth=d, g2_forest=f, g2_filter=filt, g2_labels=lab)
    z["g2_order_sensitive"] = np.int64(oc.order_sensitive(d, f, 2, 0.5))

    # g3: 2-layer stack as LayeredDecisionForest.run drives it (decision_tree.py:233-264):
    #     layer 0 (C=4) unfiltered, layer 1 (C=3) only where layer 0 says class 3, composite via COND
    d = synth.frames(["live"], 300, 60, 84)
    f0 = synth.forest(2, 6, 4, "trained", first_tree=20)
    f1 = synth.forest(3, 7, 3, "trained", first_tree=30)
    l0 = np.full((1, 30, 42), 65535, np.uint16)
    l1 = l0.copy()
    comp = l0.copy()
    oc.eval_forest(d, f0, l0, 2, None, None, 1.0)
    oc.eval_forest(d, f1, l1, 2, l0, 3, 1.0)
    bad = oc.composite([l0[0], l1[0]], COND, comp)
    assert bad == 0
    z.update(g3_depth=d, g3_forest0=f0, g3_forest1=f1, g3_cond=COND, g3_l0=l0, g3_l1=l1, g3_comp=comp)

    # g4: single tree (evaluate_image_using_tree), pre-fill 7
    d = synth.frames(["dense", "live"], 400, 33, 47)
    t = synth.trained_like_tree(40, 7, 6)
    lab = np.full(d.shape, 7, np.uint16)
    oc.eval_tree(d, t, lab)
    z.update(g4_depth=d, g4_tree=t, g4_labels=lab)

    np.savez_compressed(OUT, **z)
    print(OUT, os.path.getsize(OUT), "bytes")
    for k in ("g1_labels", "g2_labels", "g3_l0", "g3_l1", "g3_comp", "g4_labels"):
        v, c = np.unique(z[k], return_counts=True)
        print(k, dict(zip(v.tolist(), c.tolist())))
    print("g2 order-sensitive pixels:", int(z["g2_order_sensitive"]))


if __name__ == "__main__":
    main()
