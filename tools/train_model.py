#!/usr/bin/env python3
"""Command-line trainer with the arguments of /root/reference/src/train_model.py:35-50 (no window)."""
import argparse
import importlib
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser(description="Train a classifier RDF for depth images")
    ap.add_argument("--train", required=True, type=int)
    ap.add_argument("--train_block", type=int)
    ap.add_argument("--test", required=True, type=int)
    ap.add_argument("--proposals", required=True, type=int)
    ap.add_argument("--proposals_block", required=True, type=int)
    ap.add_argument("--out_trees", required=True, type=int)
    ap.add_argument("--trees_to_try", type=int)
    ap.add_argument("--depth", required=True, type=int)
    ap.add_argument("-o", "--out", required=True)
    ap.add_argument("-d", "--data", required=True)
    a = ap.parse_args()
    ds = importlib.import_module("3d-beats_amd.dataset")
    ds.train_forest(a.data, a.train, a.test, a.proposals, a.proposals_block, a.out_trees, a.depth, a.out,
                    trees_to_try=a.trees_to_try, train_block=a.train_block)


if __name__ == "__main__":
    main()
