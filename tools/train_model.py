#!/usr/bin/env python3
"""Command-line trainer with the arguments of /root/reference/src/train_model.py:35-50 (no window).

One extra argument, --tree_seed S: candidate tree i draws its proposals from numpy.random.seed(S + i).  With it the
script can be started once per GPU,
    python -m torch.distributed.run --nproc-per-node N --master-addr 127.0.0.1 tools/train_model.py ... --tree_seed S
and the ranks share the candidate trees between them (rank 0 writes the forest); without it the candidates continue
one global RNG stream as in the reference and the script runs on one GPU."""
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
    ap.add_argument("--tree_seed", type=int)
    a = ap.parse_args()
    world, rank = int(os.environ.get("WORLD_SIZE", 1)), int(os.environ.get("RANK", 0))
    if world > 1:
        if a.tree_seed is None:
            sys.exit("several ranks need --tree_seed (independent candidate trees)")
        import torch
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dev = int(os.environ.get("LOCAL_RANK", 0)) % max(1, torch.cuda.device_count())
        torch.cuda.set_device(dev)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", dev))
    ds = importlib.import_module("3d-beats_amd.dataset")
    ds.train_forest(a.data, a.train, a.test, a.proposals, a.proposals_block, a.out_trees, a.depth,
                    a.out if rank == 0 else None, trees_to_try=a.trees_to_try, train_block=a.train_block,
                    log=print if rank == 0 else (lambda *_: None), tree_seed=a.tree_seed)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
