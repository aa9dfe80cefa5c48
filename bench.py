"""Benchmark of the gradient-matching outer step (BASELINE.json metric) on MI355X.

    python bench.py --gpus 1 --steps 100 --warmup 10
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

One step = mnist_distill.py:309-374 without evaluation / logging / export: all classes, real + synthetic forward/backward,
match_loss, backward into the synthetic xyz, SGD update.  Workload = BASELINE.json configs[1] (cfg2 of SURVEY.md section 8d):
MNIST-3D-shaped synthetic stroke digits, 10 classes x 256 real clouds, 10 synthetic clouds/class x 250 points, MEConvExp, modmse.
Classes are sharded over the ranks (class c -> rank c mod N, fixed total work => "strong" scaling); the only collective is the
all-reduce of the scalar matching loss.  Rank 0 prints ONE JSON line.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

NUM_CLASSES, REAL_PER_CLASS, IPC, NUM_POINTS, POOL_PER_CLASS = 10, 256, 10, 250, 512
METRIC = "gradient-matching outer steps/sec (MNIST-3D, 10 cls)"


class KernelTimer:
    """HIP-event timing of one C-ABI entry point on the stream it is launched on (torch's current stream)."""

    def __init__(self, preds):
        self.preds, self.records, self.on = preds, [], False   # entry point -> predicate over its arguments

    def install(self):
        from dd3d_amd import _lib as L
        orig = L.call
        timer = self

        def call(name, *args):
            if timer.on and name in timer.preds and timer.preds[name](args):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                orig(name, *args)
                e1.record()
                timer.records.append((name, e0, e1, args))
            else:
                orig(name, *args)

        L.call = call  # every module calls through dd3d_amd._lib.call


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--ipc", type=int, default=IPC)
    ap.add_argument("--metric-name", default="modmse", help="dist_opt: modmse (INI default) | ours | mse | cos")
    ap.add_argument("--config", default="cfg2", choices=["cfg2", "cfg3", "cfg4", "cfg5"],
                    help="BASELINE.json configs: cfg2 (default, the metric's configuration), cfg3 = cfg2 with 50 clouds/class, "
                         "cfg4 = 40 classes x 256 real ellipsoid-shell clouds x 1024 pts, voxel 0.05, ipc 10; cfg5 = 55 classes x 256 x 2048 pts, voxel 0.02, "
                         "dense extent 26^3, in chunks of --chunk classes in sequence (record runs, not the headline)")
    ap.add_argument("--chunk", type=int, default=11, help="cfg5: classes per chunk (dd3d_amd.fused_step.ChunkedDistiller)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--emulate-shard", default="", help="R/W: time rank R's share of a W-rank job on this one GPU (no process group)")
    ap.add_argument("--real-split", default="off", help="second sharding level (graph / fused path): 'off' = classes only; 'auto' or P = the real batch "
                                                        "of every class of a shard split over P ranks, partial gradients all-reduced in the group "
                                                        "(sharding.split_layout).  With --emulate-shard the exchange is left out (compute share only)")
    ap.add_argument("--reduce-loss", default="step", choices=["step", "end"],
                    help="world > 1: when the per-rank matching losses are summed over RCCL.  step (default): after every step -- the reference "
                         "sums the classes' terms and READS the total every iteration (mnist_distill.py:361,373: loss.item() feeds loss_avg and "
                         "the least-loss checkpoint), so the per-step 4-byte all-reduce is part of the step; end: once, after the last timed "
                         "step, inside the timed region (a measurement option: config.loss_all_reduce records which one was timed)")
    ap.add_argument("--defer-loss", action="store_true", help="graph path: compute each step's loss VALUE next to the following step (measured slower: DESIGN.md section 3)")
    ap.add_argument("--path", default="graph", choices=["graph", "fused", "dropin"],
                    help="graph: class-batched step captured into one HIP graph (default); fused: the same step enqueued eagerly; "
                         "dropin: the reference-shaped autograd loop")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    local %= max(torch.cuda.device_count(), 1)   # a launcher that shows every rank only its own GPU numbers it 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1 or os.environ.get("DD3D_FORCE_DIST"):   # DD3D_FORCE_DIST: rehearse the RCCL path with a single rank
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("nccl", device_id=dev, rank=rank, world_size=world)

    from dd3d_amd.mnist_distill import Distiller
    from dd3d_amd.fused_step import FusedDistiller
    from dd3d_amd.models.MEConv import MEConvExp
    from dd3d_amd.synthetic import init_synthetic, mnist3d_like
    from dd3d_amd import ops as _ops

    torch.manual_seed(0)
    from dd3d_amd.sharding import all_reduce_in, max_over_ranks, owned_classes, reduce_loss, split_layout
    n_classes, n_points, extra = NUM_CLASSES, NUM_POINTS, {}
    if args.config == "cfg3":
        args.ipc = 50
    if args.config == "cfg5":
        from dd3d_amd.synthetic import shell_clouds
        n_classes, n_points = 55, 2048
        images_all = None
        indices_class = [list(range(c * 320, (c + 1) * 320)) for c in range(n_classes)]
        net = MEConvExp(in_channel=3, out_channel=n_classes, num_classes=n_classes, dense_extent=(26, 26, 26))
        g = torch.Generator().manual_seed(0)
        syn = (torch.rand((n_classes * args.ipc, n_points, 3), generator=g) * 1.8 - 0.9).float()
        extra = dict(real_clouds=shell_clouds(n_classes * 320, n_points, seed=0), quantization_size=0.02, clamp_range=(-0.95, 0.95),
                     zero_first_axis=False, dist_lr=1e-6)
        args.no_cpu_baseline = True
        if args.path == "graph":
            args.path = "fused"   # chunks in sequence share the allocator's memory: eager launches (GPU-bound at this size)
    elif args.config == "cfg4":
        from dd3d_amd.synthetic import shell_clouds
        n_classes, n_points = 40, 1024
        images_all = None
        indices_class = [list(range(c * 320, (c + 1) * 320)) for c in range(n_classes)]
        net = MEConvExp(in_channel=3, out_channel=n_classes, num_classes=n_classes, dense_extent=(12, 12, 12))
        g = torch.Generator().manual_seed(0)
        syn = (torch.rand((n_classes * args.ipc, n_points, 3), generator=g) * 1.8 - 0.9).float()
        extra = dict(real_clouds=shell_clouds(n_classes * 320, n_points, seed=0), quantization_size=0.05, clamp_range=(-0.95, 0.95),
                     zero_first_axis=False, dist_lr=1e-6)
        args.no_cpu_baseline = True
    else:
        images_all, _labels, indices_class = mnist3d_like(NUM_CLASSES, POOL_PER_CLASS, seed=0)
        net = MEConvExp(in_channel=3, out_channel=10)
        syn = init_synthetic(NUM_CLASSES, args.ipc, NUM_POINTS, seed=0)
    my_classes = owned_classes(n_classes, rank, world)
    er, ew = rank, world
    if args.emulate_shard:
        er, ew = (int(x) for x in args.emulate_shard.split("/"))
        my_classes = owned_classes(n_classes, er, ew)
    split_kw = {}
    if args.real_split != "off" and args.path != "dropin":
        lay = split_layout(n_classes, er, ew, 0 if args.real_split == "auto" else int(args.real_split))
        my_classes = lay["classes"]
        if lay["real_part"][1] > 1:
            if args.emulate_shard:
                exchange = lambda flat: None       # the compute share of one rank; the all-reduce is not emulated
            else:
                groups = {}                        # every rank creates every group, in the same order
                for r in range(world):
                    gr = tuple(split_layout(n_classes, r, world, lay["real_part"][1])["group_ranks"])
                    if gr not in groups:
                        groups[gr] = torch.distributed.new_group(list(gr))
                exchange = all_reduce_in(groups[tuple(lay["group_ranks"])])
            split_kw = dict(real_classes=lay["real_classes"], real_part=lay["real_part"], exchange=exchange)
    kw = dict(ipc=args.ipc, num_classes=n_classes, num_points=n_points, batch_size=REAL_PER_CLASS, dist_opt=args.metric_name, device=dev,
              seed=1234, image_syn=syn, classes=my_classes, **extra)
    if args.path == "dropin":
        d = Distiller(images_all, indices_class, net, **kw)
    elif args.config == "cfg5":
        from dd3d_amd.fused_step import ChunkedDistiller
        d = ChunkedDistiller(images_all, indices_class, net, chunk=args.chunk, **kw)
    else:
        d = FusedDistiller(images_all, indices_class, net, graph=(args.path == "graph"), defer_loss=(args.path == "graph" and args.defer_loss), **kw,
                           **split_kw)

    # ---- live timing of the dominant kernel: the Cin=128 sparse convolution forward (fp32 MFMA) ----
    BIG = 65536   # rows from which the 128-channel launches run k_conv128_rows<4> (dd3d_conv_schedule_tile_rows)
    timer = KernelTimer({"dd3d_conv_fwd": lambda a: True,                               # Cin = 128 launches and the first layer (Cin = 3)
                         "dd3d_conv128_ordered": lambda a: True,                        # forward (a[8] == 0) and mirrored = backward-data launch
                         "dd3d_conv_bwd_data": lambda a: a[5] == 128,
                         "dd3d_inorm_arp_fwd": lambda a: True,
                         "dd3d_inorm_arp_fwd_y": lambda a: True,                        # (the step's launch: no normalised rows in memory)
                         "dd3d_arp_inorm_bwd": lambda a: a[18] is not None,             # the first-order launches (they also reduce the affine gradients)
                         "dd3d_arp_inorm_bwd_y": lambda a: a[19] is not None})
    nbr_keep = []
    orig_apply_fwd = _ops.SparseConvFn.forward

    def fwd_keep(ctx, X, W, nbr, ld, M_out):
        if timer.on and W.shape[1] == 128:
            nbr_keep.append((nbr, M_out))
        return orig_apply_fwd(ctx, X, W, nbr, ld, M_out)

    _ops.SparseConvFn.forward = staticmethod(fwd_keep)
    timer.install()

    def sync_all():
        if torch.distributed.is_initialized():
            torch.distributed.barrier()
        torch.cuda.synchronize()

    pairs = []

    def one_step(**kw):
        loss = d.outer_step(**kw)
        if timer.on and hasattr(d, "pairs"):      # chunks of classes in sequence: one merged launch per chunk
            pairs.extend(p1 for (_, p1) in d.pairs)
        elif timer.on and args.path != "dropin":
            # pairs of the Cin=128 kernel maps of this step (a device->host read: only done in the kernel-timing region below)
            for i, (nbr, M, cnt) in enumerate(d.last["conv1_maps"]):
                m = M if cnt is None else d.row_counts()[i][1]
                pairs.append(int((nbr[:, :m] >= 0).sum().item()))
        # summed matching loss (mnist_distill.py:361,373): 4 bytes over RCCL when world > 1, per step or where the value is read (--reduce-loss)
        return reduce_loss(loss) if args.reduce_loss == "step" else loss

    for _ in range(max(args.warmup, 2 if args.path == "graph" else 0)):   # graph mode: step 1 sizes the workspaces, step 2 captures
        one_step()
    sync_all()
    timer.on = args.path != "graph"
    if hasattr(d, "count_pairs"):
        d.count_pairs = timer.on
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = one_step()
    if getattr(d, "defer", False):
        # deferred-loss mode: every replay carried the loss-value work of the step before it; the last step's is completed here, INSIDE the
        # timed region, so the K timed steps contain K times the whole work of a step
        loss = d.flush()
        if args.reduce_loss == "step":
            loss = reduce_loss(loss)
    if args.reduce_loss == "end":
        loss = reduce_loss(loss)   # the value is read below: summed over the ranks here, inside the timed region
    sync_all()
    dt = time.perf_counter() - t0
    timer.on = False
    dt = max_over_ranks(dt, dev)
    loss_value = float(loss.item())
    if args.path == "graph":
        # A graph replay has no per-kernel host call to bracket, so the dominant kernel is timed (HIP events on its launching stream)
        # over the SAME number of further steps, enqueued eagerly: identical kernels, arguments and launch bounds as inside the graph.
        d.check()
        timer.on = True
        for _ in range(args.steps):
            one_step(force_eager=True)
        sync_all()
        timer.on = False

    if rank == 0:
        rec = [(name, e0.elapsed_time(e1), a) for (name, e0, e1, a) in timer.records]
        # (rows, mirrored?, ms) of every Cin = 128 launch; the roofline kernel's launches are those over the merged batch (the largest row
        # bound): one forward and one mirrored (backward-data) launch per step.  The second-order launch over the synthetic rows is smaller.
        conv = []
        for (n, t, a) in rec:
            if n == "dd3d_conv128_ordered":
                conv.append((int(a[6]), bool(a[8]), t, True))
            elif n == "dd3d_conv_fwd" and a[2] == 128:
                conv.append((int(a[8]), bool(a[10]), t, False))
            elif n == "dd3d_conv_bwd_data":
                conv.append((int(a[1]), True, t, False))
        Mf = max([c[0] for c in conv], default=0)
        conv = [c for c in conv if c[0] == Mf]
        fwd, mir = [c[2] for c in conv if not c[1]], [c[2] for c in conv if c[1]]
        # The roofline figure is the FORWARD launch: it runs alone on the GPU, eagerly and inside the replayed graph alike (k_conv128_rows<4, 0> in
        # a kernel trace).  The mirrored = backward-data launch (k_conv128_rows<4, 1>, same rows and pairs) shares the matrix pipe with the
        # 128-channel kernel gradient inside the graph on purpose; its durations are reported beside it.
        ms = fwd if fwd else mir
        pairs += [int((nbr[:, :M] >= 0).sum().item()) for (nbr, M) in nbr_keep]
        avg_pairs = sum(pairs) / max(len(pairs), 1)
        avg_ms = sum(ms) / max(len(ms), 1)
        avg_flop = 2.0 * avg_pairs * 128 * 128
        mean = lambda v: sum(v) / max(len(v), 1)
        scheduled = any(c[3] for c in conv)
        tile_rows = 128 if Mf >= BIG else 32
        kname = ("k_conv128_rows<4, 0>" if tile_rows == 128 else "k_conv128<32,1,4>") + (
            " (Cin=Cout=128 convolution on the merged batch, forward launch; the mirrored = backward-data launch is listed beside it, " +
            ("scheduled launch dd3d_conv128_ordered" if scheduled else "plain launch dd3d_conv_fwd") + ", fp32 MFMA 32x32x2)")
        roof = {"bound": "mfma", "kernel": kname, "achieved": avg_flop / (avg_ms * 1e-3) / 1e12 if ms else None,
                "peak": 157.3, "unit": "TFLOP/s", "traffic": None, "traffic_measured_in_run": False, "launches": len(ms), "avg_launch_us": avg_ms * 1e3,
                "avg_launch_us_forward": mean(fwd) * 1e3, "avg_launch_us_mirrored": mean(mir) * 1e3, "avg_pairs_per_launch": avg_pairs,
                "timing": "HIP events on the launching stream around every forward launch of the kernel, over the steps enqueued eagerly after the timed "
                          "graph replays (identical kernels, arguments and launch bounds)"}
        roof["frac"] = roof["achieved"] / roof["peak"] if ms else None
        # HBM bytes per launch of this kernel: PMC counters cannot be read from inside the process, so the figure comes from the
        # committed rocprofv3 --pmc passes of this same command (tools/pmc.sh -> tools/pmc_to_json.py), corrected as the file says;
        # only attached to the run it was measured on (one GPU, all classes, default ipc)
        pmc = os.path.join(ROOT, "profiles", "r03_pmc_conv128_rows.json")
        trace = os.path.join(ROOT, "profiles", "r03_graph_step_kernel_stats.csv")
        if args.path != "dropin" and args.config == "cfg2" and args.ipc == IPC and world == 1 and not args.emulate_shard and tile_rows == 128 and os.path.exists(pmc):
            roof["traffic"] = json.load(open(pmc))["traffic_bytes_per_launch"]
            roof["traffic_source"] = "profiles/r03_pmc_conv128_rows.json (rocprofv3 --pmc passes of this command, not this run)"
            if os.path.exists(trace):
                # the same kernel INSIDE the replayed graph (next to the side streams' kernels): average duration over every launch of the
                # committed rocprofv3 --kernel-trace --stats run of this command, beside the eager-timed figure above
                import csv
                for r in csv.DictReader(open(trace)):
                    for tag, key in (("k_conv128_rows<4, 0>", "in_graph_trace"), ("k_conv128_rows<4, 1>", "in_graph_trace_mirrored")):
                        if tag in r["Name"]:
                            us = float(r["AverageNs"]) / 1e3
                            roof[key] = {"avg_launch_us": us, "achieved": avg_flop / (us * 1e-6) / 1e12, "frac": avg_flop / (us * 1e-6) / 1e12 / 157.3,
                                         "calls": int(r["Calls"]), "source": "profiles/r03_graph_step_kernel_stats.csv (not this run)"}
        step_work = None
        if args.path != "dropin":
            # the counts the algorithmic work of a step follows from (merged batch of the last step of this rank) and the
            # whole-step rates: conv0 / conv1 / classifier each run forward, backward-data and backward-weight over the merged batch
            # (the second-order pass over the synthetic prefix adds < 4 % and is left out)
            ps = d.last["passes"][0]
            M0, M1, M2 = d.row_counts()[0]
            if hasattr(d, "pairs"):   # summed over the chunks of the last step
                P0, P1 = sum(a for a, _ in d.pairs), sum(b for _, b in d.pairs)
            else:
                P0 = int((ps.maps[0][:, :M0] >= 0).sum().item())
                P1 = int((ps.maps[4][:, :M1] >= 0).sum().item())
            ncls = len(d.classes)
            flop = 3 * (2.0 * P0 * 3 * 128 + 2.0 * P1 * 128 * 128 + 2.0 * M2 * 128 * n_classes)
            step_s = dt / args.steps
            # BASELINE.json's metric also asks for the achieved HBM GB/s of the bandwidth kernels: algorithmic bytes over the
            # same event timings, per kernel family, averaged over the eagerly enqueued steps
            nch = len(d.chunks) if hasattr(d, "chunks") else 1   # launches per step of every kernel (chunks of about equal size)
            m0, m1, m2, p0 = M0 / nch, M1 / nch, M2 / nch, P0 / nch
            fam = {}
            def add(key, t_ms, nbytes):
                f = fam.setdefault(key, [0.0, 0.0])
                f[0] += t_ms
                f[1] += nbytes
            lv = 0
            for (n, t, a) in rec:
                if n in ("dd3d_inorm_arp_fwd", "dd3d_inorm_arp_fwd_y"):
                    mi, mo = ((m0, m1), (m1, m2))[lv % 2]
                    # Y read (+ xhat written by the stored-xhat launch), pooled + arg written
                    add("instance norm + affine + ReLU + max-pool forward (k_inorm_arp_fwd)", t, (512.0 if n.endswith("_y") else 1024.0) * mi + 1024.0 * mo)
                    lv += 1
                elif n == "dd3d_conv_fwd" and a[2] == 3:
                    add("first-layer convolution forward (k_conv3_fwd)", t, 4.0 * (3 * m0 + 128 * m0) + 8.0 * p0 + 4 * 27 * 3 * 128)
            lv = 0
            for (n, t, a) in rec:
                if n in ("dd3d_arp_inorm_bwd", "dd3d_arp_inorm_bwd_y"):
                    mi, mo = ((m1, m2), (m0, m1))[lv % 2]
                    add("pool + ReLU + affine + instance norm backward (k_arp_inorm_bwd)", t, 1024.0 * mi + 1024.0 * mo)
                    lv += 1
            hbm = {k: {"GB_per_s": v[1] / (v[0] * 1e-3) / 1e9, "frac_of_8TBps": v[1] / (v[0] * 1e-3) / 8e12} for k, v in fam.items() if v[0] > 0}
            step_work = {"N_bound": int(ps.N), "M0": M0, "M1": M1, "M2": M2, "P0": P0, "P1": P1, "classes_on_rank": ncls, "flop_algorithmic": flop,
                         "formula": "3*(2*P0*3*128 + 2*P1*128*128 + 2*M2*128*n_cls)", "tflops_whole_step": flop / step_s / 1e12}
        out = {"metric": METRIC, "value": args.steps / dt, "unit": "steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
               "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32",
               "data": "synthetic", "config": {"workload": (f"{args.config}: MNIST-3D-shaped stroke digits, {NUM_CLASSES} classes x {REAL_PER_CLASS} real clouds, "
                                                            f"ipc={args.ipc} x {NUM_POINTS} pts, MEConvExp, dist_opt={args.metric_name}") if args.config not in ("cfg4", "cfg5") else
                                                           (f"cfg4: ModelNet40-shaped ellipsoid shells, 40 classes x {REAL_PER_CLASS} real clouds x 1024 pts, voxel 0.05, "
                                                            f"ipc={args.ipc}, MEConvExp(num_classes=40, dense_extent=12^3), dist_opt={args.metric_name}") if args.config == "cfg4" else
                                                           (f"cfg5: ShapeNet-shaped ellipsoid shells, 55 classes x {REAL_PER_CLASS} real clouds x 2048 pts, voxel 0.02, ipc={args.ipc}, "
                                                            f"MEConvExp(num_classes=55, dense_extent=26^3), dist_opt={args.metric_name}; chunks of {args.chunk} classes in sequence"),
                                               "loss_all_reduce": ("per step" if args.reduce_loss == "step" else "once, after the last timed step (inside the timed region)") if world > 1 else None,
                                               "parallelism": f"class-sharded x{world}" + (f", real batches split x{split_kw['real_part'][1]}" if split_kw else ""),
                                               "path": {"graph": "class-batched step, one HIP graph replay per step", "fused": "class-batched step, eager launches",
                                                        "dropin": "drop-in autograd loop"}[args.path]},
               "loss": loss_value, "roofline": roof, "step_work": step_work, "hbm": hbm if args.path != "dropin" else None}
        if args.config in ("cfg4", "cfg5"):
            out["peak_hbm_GB"] = torch.cuda.max_memory_allocated() / 1e9
        if not args.no_cpu_baseline and world == 1:
            # the reference-shaped loop (the Distiller of dd3d_amd/mnist_distill.py: per-class autograd, what "loop body unchanged" means in
            # INTEGRATION.md) on the same workload, 3 steps after 1 warm-up, beside the class-batched graph step that `value` reports
            dd = Distiller(images_all, indices_class, net, **kw)
            dd.outer_step()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(3):
                dd.outer_step()
            torch.cuda.synchronize()
            out["dropin_ms_per_step"] = (time.perf_counter() - t0) / 3 * 1e3
            out["cpu_baseline"] = cpu_baseline(images_all, indices_class, net, syn, args)
            out["cpu_baseline_torch"] = cpu_baseline_torch(images_all, indices_class, net, syn, args)
            out["cpu_baseline_pytorch"] = cpu_baseline_pytorch(images_all, indices_class, args)
        print(json.dumps(out), flush=True)
    if torch.distributed.is_initialized():
        torch.distributed.destroy_process_group()


def cpu_baseline(images_all, indices_class, net, syn, args):
    """The oracle's C++/OpenMP leg (oracle/cpu/dd3d_cpu.cpp: hash-table maps, register-blocked gather convolutions, the second-order pass
    written out by hand; fp32, every host thread OpenMP gives it) on the SAME workload -- all 10 classes, 256 real clouds each, the
    reference's per-class loop, clamp and SGD with momentum on the host: one untimed warm-up step, then a bounded number of timed steps."""
    import numpy as np
    from oracle import cpu_leg as CL
    cnet = CL.Network(net, np.float32)
    rng = np.random.RandomState(1234)
    x = syn.detach().cpu().double().numpy().copy()
    mom = np.zeros_like(x)
    imgs = images_all.cpu().numpy()
    nsteps, t = 64, 0.0
    for it in range(nsteps + 1):
        reals = [imgs[rng.permutation(indices_class[c])[:REAL_PER_CLASS]] for c in range(NUM_CLASSES)]
        t0 = time.perf_counter()
        np.clip(x, 0.0, 27.0, out=x)          # mnist_distill.py:311-317
        x[:, :, 0] = 0.0
        g = CL.matching_step(cnet, x, reals, args.ipc, args.metric_name, num_points=NUM_POINTS)["grad"]
        mom = 0.5 * mom + g                    # torch.optim.SGD(lr=0.1, momentum=0.5), mnist_distill.py:268
        x -= 0.1 * mom
        if it > 0:
            t += time.perf_counter() - t0
    return {"value": nsteps / t, "unit": "steps/s", "cores": int(CL.lib().dd3d_cpu_max_threads()), "kind": "port",
            "implementation": "C++/OpenMP (oracle/cpu/dd3d_cpu.cpp), fp32",
            "sample": f"{nsteps} full steps of the same workload after 1 warm-up step, {t:.1f} s of CPU work"}


def cpu_baseline_torch(images_all, indices_class, net, syn, args):
    """The CPU oracle (torch restatement of the reference's step over the MinkowskiEngine restatement, oracle/) on the host cores:
    the SAME workload (all 10 classes), one untimed warm-up step then 8 timed steps (about 10-30 s of CPU work)."""
    from oracle import ref_model as R
    import numpy as np
    onet = R.MEConvExp(in_channel=3, out_channel=10)
    onet.load_state_dict({k: v.detach().cpu() for k, v in net.state_dict().items()})
    rng = np.random.RandomState(1234)
    osyn = syn.clone()
    osyn.requires_grad_()
    opt = torch.optim.SGD([osyn], lr=0.1, momentum=0.5)
    nsteps, t = 6, 0.0
    for it in range(nsteps + 1):
        reals = [images_all[rng.permutation(indices_class[c])[:REAL_PER_CLASS]] for c in range(NUM_CLASSES)]
        osyn.requires_grad_(False)
        R.clamp_synthetic_(osyn)
        osyn.requires_grad_()
        t0 = time.perf_counter()
        R.matching_step(onet, osyn, reals, args.ipc, args.metric_name, False, optimizer=opt)
        if it > 0:
            t += time.perf_counter() - t0
    return {"value": nsteps / t, "unit": "steps/s", "cores": torch.get_num_threads(), "kind": "port",
            "implementation": "torch (oracle/ref_model.py over oracle/minkowski_ref.py), fp32",
            "sample": f"{nsteps} full steps of the same workload after 1 warm-up step, {t:.1f} s of CPU work"}


def cpu_baseline_pytorch(images_all, indices_class, args):
    """BASELINE.json configs[0] on the host cores: the reference's ``--pytorch`` path (MEPytorch: sparse() -> dense -> Conv3d stack, full
    second order; models/MEConv.py:166-207) as restated in oracle/ref_model.py, 10 classes x 256 real clouds, 1 synthetic cloud per class.
    One untimed warm-up step, then 6 timed steps (SURVEY.md section 6 measured 9.5-20.7 s/step for the reference's own file on 8 cores)."""
    from oracle import ref_model as R
    import numpy as np
    from dd3d_amd.synthetic import init_synthetic
    torch.manual_seed(0)
    onet = R.MEPytorch(3, 10)
    rng = np.random.RandomState(1234)
    osyn = init_synthetic(NUM_CLASSES, 1, NUM_POINTS, seed=0)
    osyn.requires_grad_()
    opt = torch.optim.SGD([osyn], lr=0.1, momentum=0.5)
    nsteps, t = 6, 0.0
    for it in range(nsteps + 1):
        reals = [images_all[rng.permutation(indices_class[c])[:REAL_PER_CLASS]] for c in range(NUM_CLASSES)]
        osyn.requires_grad_(False)
        R.clamp_synthetic_(osyn)
        osyn.requires_grad_()
        t0 = time.perf_counter()
        R.matching_step(onet, osyn, reals, 1, args.metric_name, True, optimizer=opt)
        if it > 0:
            t += time.perf_counter() - t0
    return {"value": nsteps / t, "unit": "steps/s", "cores": torch.get_num_threads(), "kind": "port", "workload": "cfg1: MEPytorch (--pytorch), ipc=1",
            "sample": f"{nsteps} full steps after 1 warm-up step, {t:.1f} s of CPU work"}


if __name__ == "__main__":
    main()
