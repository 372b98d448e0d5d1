#!/usr/bin/env python3
"""bench.py -- classified Mpix/s of the RDF inference path on MI355X.

One "step" = one pass of the forest kernel over this rank's batch of synthetic 848x480 depth frames
(4 trees, depth 20, 4 classes -- BASELINE.json's metric config), frames already resident in HBM.
Default workload: 128 frames per GPU per step (= config 4's shard, 1024 frames / 8 GPUs; it is
config 2's frame x 128, half dense / half live-like), weak scaling: N GPUs evaluate N x 128 frames
and every rank's label maps reach rank 0 over xGMI inside the timed region (copy-engine peer copies into rank 0's
IPC-mapped buffer by default, an RCCL gather as the fallback: DESIGN.md section 6).  Config 2 itself
(ONE 848x480 frame per launch) is measured in the same run and reported as `cfg2_single_frame`.

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Rank 0 prints ONE JSON line.  `cpu_baseline` is this repo's CPU restatement (oracle/rdf_oracle.c,
OpenMP) -- the reference has no CPU path -- timed on a bounded sample of the same frames.
"""
import argparse
import importlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--frames", type=int, default=128, help="frames per GPU per step")
    ap.add_argument("--height", type=int, default=480)
    ap.add_argument("--width", type=int, default=848)
    ap.add_argument("--trees", type=int, default=4)
    ap.add_argument("--depth", type=int, default=20)
    ap.add_argument("--classes", type=int, default=4)
    ap.add_argument("--topology", default="full", choices=["full", "trained"])
    ap.add_argument("--chunks", type=int, default=0, help="eval/gather pipeline chunks per step (0: one launch per step; at N>1 the gather then overlaps the NEXT step)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="budget of the CPU baseline sample")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--pcie", action="store_true", help="also time the host-buffer variant (pinned H2D + kernel + D2H)")
    ap.add_argument("--headline-only", action="store_true", help="only the timed batch (no cfg2/cfg3/PCIe/CPU legs): "
                    "what tools/profile.sh traces so that rocprofv3's per-kernel average is the headline kernel's")
    ap.add_argument("--unpacked", action="store_true", help="evaluate straight from the reference-layout forest")
    ap.add_argument("--scheduler", default="dynamic", choices=["dynamic", "static", "tile"],
                    help="tile schedule of the forest kernel: persistent workgroups on a device-side queue (default), "
                         "persistent with static striding, or one workgroup per tile (non-persistent: 8 %% slower alone, "
                         "but a concurrent RCCL kernel never waits for a free slot)")
    ap.add_argument("--gather", default="auto", choices=["auto", "p2p", "rccl"],
                    help="how the label maps reach rank 0 at N>1: p2p = every rank copies its shard into rank 0's "
                         "IPC-mapped buffer with the copy engines (no CU involved); rccl = torch.distributed.gather on a "
                         "compute stream that leaves 32 CUs to RCCL; auto = p2p if the buffer can be mapped, else rccl")
    ap.add_argument("--reserve-cus", type=int, default=-1, help="run the forest kernel on a stream that leaves this many CUs "
                    "(one per shader engine = 32) to RCCL's kernels; -1: 32 at N>1, 0 at N=1 (DESIGN.md section 6)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend at N>1 (nccl = RCCL; gloo only to "
                    "rehearse the control flow with several ranks on one GPU)")
    return ap.parse_args()


class Events:
    """hipEvent pairs recorded on the stream the kernels are launched on (through the C ABI)."""

    def __init__(self, rt, n):
        import ctypes
        self.rt, self.lib, self.ct = rt, rt.lib, ctypes
        self.ev = []
        for _ in range(n):
            e = ctypes.c_void_p()
            assert self.lib.rdf_event_create(ctypes.byref(e)) == 0
            self.ev.append(e)

    def record(self, i):
        assert self.lib.rdf_event_record(self.ev[i], self.rt.stream()) == 0

    def elapsed_ms(self, i, j):
        ms = self.ct.c_float()
        assert self.lib.rdf_event_synchronize(self.ev[j]) == 0
        assert self.lib.rdf_event_elapsed_ms(self.ev[i], self.ev[j], self.ct.byref(ms)) == 0
        return float(ms.value)

    def destroy(self):
        for e in self.ev:
            self.lib.rdf_event_destroy(e)


def host_cores():
    """CPU cores this process may really use: affinity mask capped by the cgroup CPU quota."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return max(1, n)


def load_traffic(key):
    """HBM bytes per launch from a committed PMC run (profiles/roofline_traffic.json), or None."""
    p = os.path.join(ROOT, "profiles", "roofline_traffic.json")
    try:
        return json.load(open(p)).get(key, {}).get("hbm_bytes_per_launch")
    except Exception:
        return None


def main():
    a = parse()
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC (RCCL, peer-mapped buffers); before HIP starts
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", 0))
    local_rank = int(os.environ.get("LOCAL_RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    if a.gpus != world:
        if world == 1 and a.gpus > 1:
            sys.exit("bench.py --gpus N (N>1) must be launched with torch.distributed.run --nproc-per-node N")
        a.gpus = world
    dev_index = local_rank % max(1, torch.cuda.device_count())
    torch.cuda.set_device(dev_index)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if a.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", dev_index))
        else:
            dist.init_process_group(a.backend, rank=rank, world_size=world)

    rdf = importlib.import_module("3d-beats_amd")
    dmod = importlib.import_module("3d-beats_amd.distributed")
    synth = rdf.synth
    rt = rdf.get_runtime()  # raises without the HIP library or a device: no CPU fallback
    lib = rt.lib

    H, W, F, T, D, C = a.height, a.width, a.frames, a.trees, a.depth, a.classes
    forest_np = synth.forest(T, D, C, a.topology)
    forest = rdf.DecisionForest.from_numpy(forest_np)
    frames_np = synth.mixed_batch(F, first_idx=rank * F, h=H, w=W)
    depth = rdf.to_device(frames_np)
    labels = rdf.DeviceArray((F, H, W), np.uint16).fill(65535)
    ev = rdf.DecisionTreeEvaluator(use_packed=not a.unpacked)
    lib.rdf_set_scheduler({"dynamic": 1, "static": 0, "tile": 2}[a.scheduler])
    if not a.unpacked:
        forest.packed(1.0)  # load-time repack, outside the timed region (like the reference's upload)
    # N>1: one launch per step; the gather of step s overlaps the evaluation of step s+1 (two label buffers);
    # --chunks C > 0 selects the in-step pipeline instead (C launches, gather of chunk c overlaps chunk c+1)
    chunks = a.chunks or 1
    overlapped = world > 1 and a.chunks == 0
    sharded = dmod.ShardedForestEvaluator(ev, forest, F, (H, W), n_chunks=chunks)
    ring = [labels, rdf.DeviceArray((F, H, W), np.uint16).fill(65535)] if overlapped else None

    # N>1, default: peer copies over xGMI by the copy engines (RCCL carries only the control plane); needs HIP IPC
    # between the ranks' processes, falls back to the RCCL gather if any rank cannot map rank 0's buffer
    peer = None
    if world > 1 and a.chunks == 0 and a.gather in ("auto", "p2p"):
        pg = dmod.PeerCopyGather(world, rank, F * H * W * 2)
        if pg.ok:
            peer = dmod.PeerCopyForestEvaluator(ev, forest, F, (H, W), pg)
        elif a.gather == "p2p":
            sys.exit("--gather p2p: rank 0's receive buffer could not be mapped by every rank")
    gather_mode = None if world == 1 else ("p2p copy engines" if peer is not None else "rccl gather")

    def one_step():
        if peer is not None:
            peer.step(depth, ring)
        elif overlapped:
            sharded.step_overlapped(depth, ring)
        else:
            sharded.step(depth, labels)

    def drain():
        (peer if peer is not None else sharded).drain()

    # ---- algorithmic bytes of one step, from the visit counters of the same walk ----
    dstats = rdf.DeviceArray((3,), np.uint64).fill(0)
    scratch = rdf.DeviceArray((F, H, W), np.uint16).fill(65535)
    rc = lib.rdf_eval_forest_stats(depth.ptr, F, W, H, forest.forest_cu.ptr, T, D, C, None, -1, scratch.ptr, 1, 1.0,
                                   dstats.ptr, rt.stream())
    assert rc == 0, lib.rdf_error_string(rc)
    stats = dstats.get()
    alg_bytes = synth.algorithmic_bytes(F, H, W, 1, False, C, stats)

    # ---- N>1: the compute stream leaves one CU per shader engine to RCCL (its send/recv kernel cannot start beside the
    # forest kernel's persistent workgroups otherwise: tools/ubench_overlap.py) ----
    reserve = a.reserve_cus if a.reserve_cus >= 0 else (32 if world > 1 and peer is None else 0)
    compute_stream, masked_handle = None, None
    if reserve > 0:
        import ctypes
        h = ctypes.c_void_p()
        rc_m = lib.rdf_stream_create_with_reserved_cus(ctypes.byref(h), reserve)
        if rc_m == 0:
            masked_handle = h
            compute_stream = torch.cuda.ExternalStream(h.value)
        else:
            print(f"rank {rank}: no CU-masked stream ({lib.rdf_error_string(rc_m)}); using the current stream", file=sys.stderr)
            reserve = 0
    torch.cuda.synchronize()
    import contextlib
    stream_ctx = torch.cuda.stream(compute_stream) if compute_stream is not None else contextlib.nullcontext()

    # ---- timed region ----
    def sync_all():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    with stream_ctx:
        for _ in range(a.warmup):
            one_step()
        drain()
        evs = Events(rt, 2 * a.steps)
        sync_all()
        t0 = time.perf_counter()
        for i in range(a.steps):
            evs.record(2 * i)
            one_step()
            evs.record(2 * i + 1)
        drain()              # every step's label maps are on rank 0 before the clock stops
        sync_all()
        elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    kern_ms = [evs.elapsed_ms(2 * i, 2 * i + 1) for i in range(a.steps)]
    evs.destroy()
    assert np.array_equal(labels.get(), scratch.get()), "timed path and stats path disagree"

    # ---- N>1: did rank 0 really receive every rank's label maps?  (checksum of checksums) ----
    gather_check = None
    if world > 1:
        mine = labels.torch_bytes().view(torch.int16).to(torch.int64)
        sums = torch.stack([mine.sum(), (mine * (torch.arange(mine.numel(), device=mine.device) % 8191)).sum()])
        allsums = [torch.zeros_like(sums) for _ in range(world)]
        dist.all_gather(allsums, sums)
        if rank == 0:
            got = peer.result() if peer is not None else sharded.result()
            ok = True
            for g in range(world):
                part = got[g * F:(g + 1) * F]
                part = (part.reshape(-1) if peer is not None else part.torch_bytes().view(torch.int16)).to(torch.int64)
                chk = torch.stack([part.sum(), (part * (torch.arange(part.numel(), device=part.device) % 8191)).sum()])
                ok = ok and bool(torch.equal(chk, allsums[g]))
            gather_check = "ok" if ok else "MISMATCH"

    pix_per_step = world * F * H * W
    value = pix_per_step * a.steps / elapsed / 1e6
    kern_avg_s = float(np.mean(kern_ms)) / 1e3
    achieved = alg_bytes / kern_avg_s / 1e9

    out = {
        "metric": "classified Mpix/s on 848x480 depth frames (4 trees, depth 20); % HBM roofline",
        "value": round(value, 2), "unit": "Mpix/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
        "ms_per_step": round(elapsed / a.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"{F} x {W}x{H} depth frames per GPU per step (config 4 shard = config 2 frame x {F}; "
                               f"half dense, half live-like), T{T}/D{D}/C{C} {a.topology} forest, "
                               + (f"labels gathered to rank 0 ({gather_mode}; control plane {a.backend}) inside the timed region" if world > 1 else "1 GPU"),
                   "frames_per_gpu": F, "frame": [H, W], "trees": T, "tree_depth": D, "classes": C,
                   "topology": a.topology, "forest_layout": "reference" if a.unpacked else "packed16+exact32",
                   "pipeline_chunks": chunks, "tile_schedule": a.scheduler, "cus_left_to_rccl": reserve, "gather": gather_mode, "gather_overlap": ("next step" if (overlapped or peer is not None) else ("in-step chunks" if world > 1 else None)), "sharding": f"frames x{world}, forest replicated",
                   "gather_check": gather_check},
        "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": round(achieved / HBM_PEAK_GBS, 4),
                     "traffic": load_traffic(f"F{F}_T{T}_D{D}_C{C}_{a.topology}"),
                     "kernel": "k_eval_forest", "kernel_ms": round(kern_avg_s * 1e3, 4),
                     "algorithmic_bytes_per_launch": int(alg_bytes),
                     "algorithmic_bytes_per_pixel": round(alg_bytes / (F * H * W), 1),
                     "visits": {"pixels": int(stats[0]), "node_records": int(stats[1]), "leaves": int(stats[2])}},
    }

    if rank == 0 and world == 1 and not a.headline_only:
        # ---- what HBM really delivers to a plain stream: a 1-GB device-to-device copy (SURVEY 8d asks for the roofline
        # against the spec peak AND a measured copy ceiling) ----
        src_t = torch.empty(1 << 30, dtype=torch.uint8, device="cuda").fill_(1)
        dst_t = torch.empty_like(src_t)
        c0, c1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        best = None
        for _ in range(4):
            c0.record()
            dst_t.copy_(src_t)
            c1.record()
            torch.cuda.synchronize()
            ms = c0.elapsed_time(c1)
            best = ms if best is None else min(best, ms)
        ceiling = 2.0 * (1 << 30) / (best * 1e-3) / 1e9      # bytes read + bytes written
        out["roofline"]["copy_ceiling"] = {"value": round(ceiling, 1), "unit": "GB/s",
                                           "frac_of_ceiling": round(achieved / ceiling, 4),
                                           "hbm_traffic_gbs": (round(out["roofline"]["traffic"] / kern_avg_s / 1e9, 1)
                                                               if out["roofline"]["traffic"] else None),
                                           "what": "1-GB device-to-device copy, read + written bytes per second"}
        del src_t, dst_t

        # ---- config 2 proper: ONE 848x480 frame per launch (dense frame 0) ----
        one_d, one_l = depth[0:1], rdf.DeviceArray((1, H, W), np.uint16).fill(65535)
        for _ in range(10):
            ev.get_labels_forest(forest, one_d, one_l)
        n1 = 200
        e1 = Events(rt, 2)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        e1.record(0)
        for _ in range(n1):
            ev.get_labels_forest(forest, one_d, one_l)
        e1.record(1)
        torch.cuda.synchronize()
        wall1 = (time.perf_counter() - t1) / n1
        dev1 = e1.elapsed_ms(0, 1) / n1 / 1e3
        e1.destroy()
        out["cfg2_single_frame"] = {"value": round(H * W / wall1 / 1e6, 2), "unit": "Mpix/s",
                                    "ms_per_frame_wall": round(wall1 * 1e3, 4), "ms_per_frame_device": round(dev1 * 1e3, 4),
                                    "frame": "dense #0", "launches": n1}

        # ---- config 3: 2-layer stack through LayeredDecisionForest.run (run_live_layered.py:126), r = 2 ----
        cfg3 = {"layers": [{"model": forest}, {"model": forest, "filter_model": 0, "filter_model_class": 3}],
                "conditions": [[0, 1], [0, 2], [1, 3], [0, 3], [0, 4], [0, 5], [0, 6]],
                "label_colors": [[0, 0, 0, 255]] * 6}
        lf = rdf.LayeredDecisionForest(cfg3, (H, W), 2)
        dbuf, lbuf = rdf.GpuBuffer((H, W), np.uint16), rdf.GpuBuffer((H // 2, W // 2), np.uint16)
        dbuf.cu().set(frames_np[1 if F > 1 else 0])   # a live-like frame
        for _ in range(10):
            lf.run(dbuf, lbuf, 1.0)
        torch.cuda.synchronize()
        t3 = time.perf_counter()
        for _ in range(100):
            lf.run(dbuf, lbuf, 1.0)
        torch.cuda.synchronize()
        w3 = (time.perf_counter() - t3) / 100
        out["cfg3_layered_run"] = {"ms_per_frame_wall": round(w3 * 1e3, 4), "value": round(H * W / w3 / 1e6, 2),
                                   "unit": "Mpix/s", "what": "LayeredDecisionForest.run, 2 layers (second filtered on "
                                   "class 3 of the first), labels_reduce 2, one live-like 848x480 frame: 3 fills + 2 "
                                   "forest launches + composite"}

        # ---- host-buffer variant: pinned H2D of the frames + kernel + D2H of the labels (never `value`) ----
        # Opt-in: its launches carry the headline kernel's name and would blur rocprofv3's per-kernel average.
        if a.pcie:
            pin_in = torch.from_numpy(frames_np.view(np.int16).reshape(-1)).pin_memory()
            pin_out = torch.empty(F * H * W, dtype=torch.int16).pin_memory()
            d_t, l_t = depth.torch_bytes().view(torch.int16), labels.torch_bytes().view(torch.int16)
            for rep in range(4):
                if rep == 1:
                    torch.cuda.synchronize()
                    tp = time.perf_counter()
                d_t.copy_(pin_in, non_blocking=True)
                ev.get_labels_forest(forest, depth, labels)
                pin_out.copy_(l_t, non_blocking=True)
            torch.cuda.synchronize()
            wall_p = (time.perf_counter() - tp) / 3
            out["pcie_inclusive"] = {"value": round(F * H * W / wall_p / 1e6, 2), "unit": "Mpix/s",
                                     "ms_per_step": round(wall_p * 1e3, 3),
                                     "what": "pinned H2D of the batch + kernel + D2H of the labels, serial on one stream"}
            # the same work as a 3-stage pipeline over 4 chunks of the batch: upload chunk c+1 and download chunk c-1
            # (copy engines, one stream each) while chunk c is evaluated; steps follow each other without a gap
            n_ch = 4
            cuts = [(F * c) // n_ch for c in range(n_ch + 1)]
            s_up, s_dn = torch.cuda.Stream(), torch.cuda.Stream()
            cur = torch.cuda.current_stream()
            per = H * W
            up_done = [[torch.cuda.Event() for _ in range(n_ch)] for _ in range(2)]
            free_in = [None] * n_ch      # event: the kernel that read chunk c's depth slot has finished
            free_out = [None] * n_ch     # event: chunk c's labels have been downloaded
            reps = 4
            for rep in range(reps + 1):
                if rep == 1:
                    torch.cuda.synchronize()
                    tp = time.perf_counter()
                for c in range(n_ch):
                    a0, a1 = cuts[c], cuts[c + 1]
                    with torch.cuda.stream(s_up):
                        if free_in[c] is not None:
                            s_up.wait_event(free_in[c])
                        d_t[a0 * per:a1 * per].copy_(pin_in[a0 * per:a1 * per], non_blocking=True)
                        up_done[rep & 1][c].record(s_up)
                    cur.wait_event(up_done[rep & 1][c])
                    if free_out[c] is not None:
                        cur.wait_event(free_out[c])
                    ev.get_labels_forest(forest, depth[a0:a1], labels[a0:a1])
                    k_done = torch.cuda.Event()
                    k_done.record(cur)
                    free_in[c] = k_done
                    with torch.cuda.stream(s_dn):
                        s_dn.wait_event(k_done)
                        pin_out[a0 * per:a1 * per].copy_(l_t[a0 * per:a1 * per], non_blocking=True)
                        e = torch.cuda.Event()
                        e.record(s_dn)
                        free_out[c] = e
            torch.cuda.synchronize()
            wall_pp = (time.perf_counter() - tp) / reps
            same = bool(np.array_equal(pin_out.numpy().view(np.uint16).reshape(F, H, W), scratch.get()))
            out["pcie_inclusive_pipelined"] = {"value": round(F * H * W / wall_pp / 1e6, 2), "unit": "Mpix/s",
                                               "ms_per_step": round(wall_pp * 1e3, 3), "labels_match": same,
                                               "what": f"same transfers, {n_ch} chunks per step on three streams: upload, "
                                                       "evaluate and download overlap"}

        if not a.no_cpu_baseline:
            from oracle import rdf_oracle  # the checker; never the thing measured as `value`
            got = labels.get()
            cores = min(host_cores(), rdf_oracle.max_threads())
            order = list(range(F))  # the batch alternates dense / live-like frames
            done, t_cpu, mism = 0, 0.0, 0
            for i in order:
                want = np.full((1, H, W), 65535, np.uint16)
                tc = time.perf_counter()
                rdf_oracle.eval_forest(frames_np[i:i + 1], forest_np, want, n_threads=cores)
                t_cpu += time.perf_counter() - tc
                mism += int((want[0] != got[i]).sum())
                done += 1
                if t_cpu >= a.cpu_seconds:
                    break
            out["cpu_baseline"] = {"value": round(done * H * W / t_cpu / 1e6, 3), "unit": "Mpix/s", "cores": cores,
                                   "kind": "port",
                                   "sample": f"{done} of the step's {F} frames (alternating dense/live-like), "
                                             f"{t_cpu:.1f} s of oracle/rdf_oracle.c (OpenMP, -O2) on the host; "
                                             f"GPU labels of those frames differ in {mism} pixels"}
            assert mism == 0, f"GPU labels differ from the oracle in {mism} pixels"
            # what a "numpy fallback" would have been (variant ii): level-synchronous numpy, one process
            from oracle import rdf_numpy
            wn = np.full((1, H, W), 65535, np.uint16)
            tn = time.perf_counter()
            rdf_numpy.eval_forest(frames_np[1:2], forest_np, wn)
            tn = time.perf_counter() - tn
            out["cpu_baseline_numpy"] = {"value": round(H * W / tn / 1e6, 3), "unit": "Mpix/s", "cores": 1,
                                         "kind": "port", "sample": f"1 live-like frame, {tn:.1f} s of oracle/rdf_numpy.py; "
                                         f"labels differ from the GPU's in {int((wn[0] != got[1]).sum())} pixels"}

    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        if peer is not None:            # unmap on the peers before rank 0 frees the buffer
            if rank != 0:
                pg.close()
            dist.barrier()
            if rank == 0:
                pg.close()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
