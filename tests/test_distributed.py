"""N>1 path on CPU: 2 gloo ranks, host-memory test double, shards gathered on rank 0 == oracle."""
import importlib
import os
import socket
import sys

import numpy as np
import pytest


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_chunks, tmpdir):
    import torch.distributed as dist
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for p in (root, os.path.join(root, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    rdf = importlib.import_module("3d-beats_amd")
    dmod = importlib.import_module("3d-beats_amd.distributed")
    import fake_runtime
    from oracle import rdf_oracle
    rdf.set_runtime(fake_runtime.HostRuntime())
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        frames, h, w, r = 5, 36, 52, 2
        forest_np = rdf.synth.forest(3, 7, 4, "trained")
        forest = rdf.DecisionForest.from_numpy(forest_np)
        a, b = dmod.shard_range(world * frames, rank, world)
        assert b - a == frames
        mine = rdf.synth.mixed_batch(frames, first_idx=a, h=h, w=w)
        ev = rdf.DecisionTreeEvaluator()
        sh = dmod.ShardedForestEvaluator(ev, forest, frames, (h, w), labels_reduce=r, scale_factor=0.5,
                                         n_chunks=max(n_chunks, 1))
        depth = rdf.to_device(mine)
        labels = rdf.DeviceArray((frames, h // r, w // r), np.uint16)
        if n_chunks == 0:      # cross-step pipelining: one launch per step, gather overlaps the next step
            ring = [labels, rdf.DeviceArray((frames, h // r, w // r), np.uint16)]
            for _ in range(3):
                sh.step_overlapped(depth, ring, prefill=65535)
            sh.drain()
        else:
            for _ in range(2):
                sh.step(depth, labels, prefill=65535)
        dist.barrier()
        if rank == 0:
            got = sh.result().get()
            assert got.shape == (world * frames, h // r, w // r)
            for g in range(world):
                fr = rdf.synth.mixed_batch(frames, first_idx=g * frames, h=h, w=w)
                want = np.full((frames, h // r, w // r), 65535, np.uint16)
                rdf_oracle.eval_forest(fr, forest_np, want, r, scale_factor=0.5)
                assert np.array_equal(got[g * frames:(g + 1) * frames], want), f"shard {g}"
            open(os.path.join(tmpdir, "ok"), "w").write("ok")
        else:
            assert sh.result() is None
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n_chunks", [1, 3, 0])
def test_two_rank_gather_matches_oracle(n_chunks, tmp_path, oracle):
    import torch.multiprocessing as mp
    mp.spawn(_worker, args=(2, _free_port(), n_chunks, str(tmp_path)), nprocs=2, join=True)
    assert (tmp_path / "ok").exists()


def test_shard_range_partitions_the_batch(rdf):
    dmod = importlib.import_module("3d-beats_amd.distributed")
    for n, wsz in [(1024, 8), (10, 4), (3, 8), (256, 8)]:
        spans = [dmod.shard_range(n, g, wsz) for g in range(wsz)]
        assert spans[0][0] == 0 and spans[-1][1] == n
        assert all(spans[i][1] == spans[i + 1][0] for i in range(wsz - 1))
    assert dmod.shard_range(1024, 3, 8) == (384, 512)
