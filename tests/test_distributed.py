"""N>1 path on CPU: 2 gloo ranks, host-memory test double, shards gathered on rank 0 == oracle."""
import importlib
import time
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


def test_eight_rank_gather_matches_oracle(tmp_path, oracle):
    """BASELINE configs 4 and 5 run on EIGHT ranks, and no GPU box this build can lease holds more than one card (nor lets more
    than six processes touch it: the four-rank rehearsal of bench.py is the widest the GPU suite can go).  So the eight-rank
    shape of the sharded path -- shard_range over 8, rank-indexed receive buffer [8, frames, ...], eight senders in one gather,
    the cross-step overlap with two label buffers -- is rehearsed here, on the CPU, with gloo and the host double: every
    shard as rank 0 received it equals the oracle's labels for that shard's frames."""
    import torch.multiprocessing as mp
    mp.spawn(_worker, args=(8, _free_port(), 0, str(tmp_path)), nprocs=8, join=True)
    assert (tmp_path / "ok").exists()


def _replicate_worker(rank, world, port, tmpdir):
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
        frames, h, w = 3, 36, 52
        forest_np = rdf.synth.forest(3, 7, 4, "trained", 50)
        # only rank 0 has the model; the others hold an empty forest of the same shape
        forest = rdf.DecisionForest.from_numpy(forest_np) if rank == 0 else rdf.DecisionForest(3, 7, 4)
        ev = rdf.DecisionTreeEvaluator()
        mine = rdf.synth.mixed_batch(frames, first_idx=rank * frames, h=h, w=w)
        depth, labels = rdf.to_device(mine), rdf.DeviceArray((frames, h, w), np.uint16).fill(65535)
        if rank != 0:
            ev.get_labels_forest(forest, depth, labels)          # (packs the EMPTY forest: the replica must not reuse that table)
            labels.fill(65535)
        assert dmod.replicate_forest(forest, src=0) is forest
        assert np.array_equal(forest.forest_cu.get(), forest_np)
        ev.get_labels_forest(forest, depth, labels)
        want = np.full((frames, h, w), 65535, np.uint16)
        rdf_oracle.eval_forest(mine, forest_np, want)
        assert np.array_equal(labels.get(), want), f"rank {rank}"
        packs = [c for c in rdf.get_runtime().lib.calls if c[0] == "rdf_forest_pack"]
        assert len(packs) == (1 if rank == 0 else 2)            # rank 1 re-packed after the broadcast changed forest_cu
        dist.barrier()
        open(os.path.join(tmpdir, f"ok{rank}"), "w").write("ok")
    finally:
        dist.destroy_process_group()


def test_forest_is_replicated_by_one_broadcast(tmp_path, oracle):
    """the forest is replicated -- rank 0 holds the model, one broadcast of forest_cu gives it to the others, whose
    packed-table cache notices the new contents; every rank's shard then gets the oracle's labels (two gloo ranks, host double)."""
    import torch.multiprocessing as mp
    mp.spawn(_replicate_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    assert (tmp_path / "ok0").exists() and (tmp_path / "ok1").exists()


def test_shard_range_partitions_the_batch(rdf):
    dmod = importlib.import_module("3d-beats_amd.distributed")
    for n, wsz in [(1024, 8), (10, 4), (3, 8), (256, 8)]:
        spans = [dmod.shard_range(n, g, wsz) for g in range(wsz)]
        assert spans[0][0] == 0 and spans[-1][1] == n
        assert all(spans[i][1] == spans[i + 1][0] for i in range(wsz - 1))
    assert dmod.shard_range(1024, 3, 8) == (384, 512)


def _p2p_worker(rank, world, port, tmpdir, multi_device=False):
    """Two processes on the one GPU of the test box: rank 0 exports its receive buffer, rank 1 maps it through HIP
    IPC, both copy their label maps in with hipMemcpyAsync on a side stream; gloo only carries the control plane.
    multi_device: rank g uses GPU g (the copy then crosses xGMI) and RCCL carries the control plane."""
    import torch
    import torch.distributed as dist
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for p in (root, os.path.join(root, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(rank if multi_device else 0)
    rdf = importlib.import_module("3d-beats_amd")
    dmod = importlib.import_module("3d-beats_amd.distributed")
    from oracle import rdf_oracle
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    if multi_device:
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", rank))
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        frames, h, w, r = 6, 120, 200, 2
        forest_np = rdf.synth.forest(4, 9, 4, "trained")
        forest = rdf.DecisionForest.from_numpy(forest_np)
        mine = rdf.synth.mixed_batch(frames, first_idx=rank * frames, h=h, w=w)
        ev = rdf.DecisionTreeEvaluator()
        nbytes = frames * (h // r) * (w // r) * 2
        # a rank that cannot map the buffer turns the mode off on EVERY rank (callers then use the RCCL gather)
        broken = dmod.PeerCopyGather(world, rank, nbytes, _fail_open_on_rank=1)
        assert not broken.ok and broken.base is None
        assert broken.errors and "hipIpcOpenMemHandle" in broken.errors[1]
        gather = dmod.PeerCopyGather(world, rank, nbytes)
        assert gather.ok, "HIP IPC between two processes on one GPU should work"
        pe = dmod.PeerCopyForestEvaluator(ev, forest, frames, (h, w), gather, labels_reduce=r, scale_factor=0.5)
        depth = rdf.to_device(mine)
        ring = [rdf.DeviceArray((frames, h // r, w // r), np.uint16) for _ in range(2)]
        for _ in range(5):
            pe.step(depth, ring, prefill=65535)
        pe.drain()
        torch.cuda.synchronize()
        dist.barrier()
        wants = {}

        def want_for(g, first):
            if (g, first) not in wants:
                fr = rdf.synth.mixed_batch(frames, first_idx=first, h=h, w=w)
                want = np.full((frames, h // r, w // r), 65535, np.uint16)
                rdf_oracle.eval_forest(fr, forest_np, want, r, scale_factor=0.5)
                wants[(g, first)] = want
            return wants[(g, first)]
        if rank == 0:
            got = pe.result().cpu().numpy().view(np.uint16)
            assert got.shape == (world * frames, h // r, w // r)
            for g in range(world):
                assert np.array_equal(got[g * frames:(g + 1) * frames], want_for(g, g * frames)), f"shard {g}"
            cnt = gather.ready_counters()
            assert (cnt[4 % 2] == 5).all() and (cnt[3 % 2] == 4).all()        # steps 4 and 3 are marked as landed
        else:
            assert pe.result() is None
        dist.barrier()

        # ---- the ring as an API: a consumer on rank 0 that is slower than the producers.  Three different batches in turn,
        # so that a slot overwritten too early (step s + 2 lands in the slot of step s) would show; the producers wait for
        # the consumer's release before they reuse a slot (flow control) ----
        gather2 = dmod.PeerCopyGather(world, rank, nbytes, n_slots=2)
        assert gather2.ok
        pe2 = dmod.PeerCopyForestEvaluator(ev, forest, frames, (h, w), gather2, labels_reduce=r, scale_factor=0.5, flow_control=True)
        batches = [rdf.to_device(rdf.synth.mixed_batch(frames, first_idx=1000 * b + rank * frames, h=h, w=w)) for b in range(3)]
        n_steps = 7
        for s_no in range(n_steps):
            pe2.step(batches[s_no % 3], ring, prefill=65535)
            if rank == 0:
                time.sleep(0.03)                                   # the consumer dawdles: the producers run ahead and must wait
                assert gather2.wait_ready(s_no, timeout_s=20.0), s_no
                got = gather2.slot_array(s_no).cpu().numpy().view(np.uint16).reshape(world * frames, h // r, w // r)
                for g in range(world):
                    assert np.array_equal(got[g * frames:(g + 1) * frames], want_for(g, 1000 * (s_no % 3) + g * frames)), (s_no, g)
                gather2.release(s_no)
        pe2.drain()
        torch.cuda.synchronize()
        dist.barrier()

        # ---- direct stores: every rank's KERNEL writes its label maps into its place in rank 0's ring (no copy, no second
        # label buffer); same dawdling consumer, same three batches ----
        gather3 = dmod.PeerCopyGather(world, rank, nbytes, n_slots=2)
        assert gather3.ok
        pe3 = dmod.PeerCopyForestEvaluator(ev, forest, frames, (h, w), gather3, labels_reduce=r, scale_factor=0.5, flow_control=True,
                                           direct_stores=True)
        for s_no in range(n_steps):
            out = pe3.step(batches[s_no % 3], None, prefill=65535)
            assert out.ptr == gather3.slot_ptr(rank, 0, s_no)             # the labels ARE the ring slot, on every rank
            if rank == 0:
                time.sleep(0.03)
                assert gather3.wait_ready(s_no, timeout_s=20.0), s_no
                got = gather3.slot_array(s_no).cpu().numpy().view(np.uint16).reshape(world * frames, h // r, w // r)
                for g in range(world):
                    assert np.array_equal(got[g * frames:(g + 1) * frames], want_for(g, 1000 * (s_no % 3) + g * frames)), ("direct", s_no, g)
                gather3.release(s_no)
        pe3.drain()
        torch.cuda.synchronize()
        if rank == 0:
            open(os.path.join(tmpdir, "ok"), "w").write("ok")
        dist.barrier()
        gather3.close()
        gather2.close()
        dist.barrier()          # rank 0 is done reading before anyone unmaps
        gather.close()
    finally:
        dist.destroy_process_group()


@pytest.mark.gpu
def test_two_process_peer_copy_gather_on_one_gpu(tmp_path, oracle):
    import torch.multiprocessing as mp
    mp.spawn(_p2p_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    assert (tmp_path / "ok").exists()


@pytest.mark.gpu
def test_peer_copy_gather_across_two_devices(tmp_path, oracle):
    """The same exchange with the two ranks on two DIFFERENT GPUs: rank 1's hipMemcpyAsync into rank 0's IPC-mapped
    buffer crosses xGMI, RCCL carries the control plane.  Needs a box with at least two GPUs (the 1-GPU test box skips)."""
    import torch
    import torch.multiprocessing as mp
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs")
    mp.spawn(_p2p_worker, args=(2, _free_port(), str(tmp_path), True), nprocs=2, join=True)
    assert (tmp_path / "ok").exists()


def _replicate_gpu_worker(rank, world, port, tmpdir):
    """Two processes on the one GPU: rank 0 packs AND tunes, one broadcast each gives rank 1 the forest and the tuned table."""
    import torch
    import torch.distributed as dist
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for p in (root, os.path.join(root, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    torch.cuda.set_device(0)
    rdf = importlib.import_module("3d-beats_amd")
    dmod = importlib.import_module("3d-beats_amd.distributed")
    from oracle import rdf_oracle
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        frames, h, w = 3, 120, 212
        forest_np = rdf.synth.forest(4, 11, 4, "balanced", 70, calib=rdf.synth.calibration_frames(3, 120, 212))
        lib = rdf.get_runtime().lib
        if rank == 0:
            forest = rdf.DecisionForest.from_numpy(forest_np)
            assert lib.rdf_forest_set_deep_from(forest.packed(1.0).ptr, 6) == 0       # (stands in for forest.tune(sample))
        else:
            forest = rdf.DecisionForest(4, 11, 4)
        dmod.replicate_forest(forest, src=0, packed_scales=(1.0,))
        assert np.array_equal(forest.forest_cu.get(), forest_np)
        table = forest._packed[1.0][1]
        assert forest.packed(1.0) is table                                           # the cache takes the replica: no re-pack
        assert forest.deep_from(1.0) == 6                                            # the choice came with the table
        mine = rdf.synth.mixed_batch(frames, first_idx=rank * frames, h=h, w=w)
        depth, labels = rdf.to_device(mine), rdf.DeviceArray((frames, h, w), np.uint16).fill(65535)
        rdf.DecisionTreeEvaluator().get_labels_forest(forest, depth, labels)
        want = np.full((frames, h, w), 65535, np.uint16)
        rdf_oracle.eval_forest(mine, forest_np, want)
        assert np.array_equal(labels.get(), want), f"rank {rank}"
        st8 = rdf.DeviceArray((8,), np.uint64).fill(0)
        assert lib.rdf_eval_forest_packed_stats(depth.ptr, frames, w, h, table.ptr, forest.forest_cu.ptr, 4, 11, 4, labels.ptr, 1, st8.ptr,
                                                rdf.get_runtime().stream()) == 0
        assert int(st8.get()[7]) > 0                                                 # ... and it is walked: deep blocks are fetched
        dist.barrier()
        open(os.path.join(tmpdir, f"ok{rank}"), "w").write("ok")
    finally:
        dist.destroy_process_group()


@pytest.mark.gpu
def test_packed_and_tuned_table_is_replicated_to_every_rank(tmp_path, oracle):
    import torch.multiprocessing as mp
    mp.spawn(_replicate_gpu_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    assert (tmp_path / "ok0").exists() and (tmp_path / "ok1").exists()


def _train_worker(rank, world, port, tmpdir):
    """Two processes on the one GPU train the candidate trees of a forest between them."""
    import torch
    import torch.distributed as dist
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for p in (root, os.path.join(root, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    torch.cuda.set_device(0)
    importlib.import_module("3d-beats_amd")
    ds_mod = importlib.import_module("3d-beats_amd.dataset")
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        forest, pct = ds_mod.train_forest(os.path.join(tmpdir, "data"), 6, 4, 32, 16, 2, 6, None, trees_to_try=5,
                                          log=lambda *_: None, tree_seed=1234)
        np.save(os.path.join(tmpdir, f"forest_rank{rank}.npy"), forest)
        open(os.path.join(tmpdir, f"pct_rank{rank}.txt"), "w").write(repr(pct))
    finally:
        dist.destroy_process_group()


@pytest.mark.gpu
def test_candidate_trees_trained_by_two_processes_give_the_single_process_forest(rdf, gpu_runtime, tmp_path):
    import torch.multiprocessing as mp
    from test_training import make_data
    ds_mod = importlib.import_module("3d-beats_amd.dataset")
    depth, labels = make_data(rdf, n=10, h=48, w=64, first=300)
    ds_mod.write_dataset(str(tmp_path / "data"), depth, labels, {1: [255, 0, 0, 255], 2: [0, 255, 0, 255], 3: [0, 0, 255, 255]})
    want, want_pct = ds_mod.train_forest(str(tmp_path / "data"), 6, 4, 32, 16, 2, 6, None, trees_to_try=5,
                                         log=lambda *_: None, tree_seed=1234)
    mp.spawn(_train_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    for r in range(2):
        got = np.load(tmp_path / f"forest_rank{r}.npy")
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), f"rank {r}"
        assert eval(open(tmp_path / f"pct_rank{r}.txt").read()) == want_pct
    # and tree_seed really decouples the candidates: a different seed gives a different forest
    other, _ = ds_mod.train_forest(str(tmp_path / "data"), 6, 4, 32, 16, 2, 6, None, trees_to_try=5,
                                   log=lambda *_: None, tree_seed=99)
    assert not np.array_equal(other, want)
