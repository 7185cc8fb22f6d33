"""The kernel-level measurement (one process per GPU): the timed region is
``whisper_timestamped.pipeline.HotPathPipeline.submit`` in a loop -- the product's scheduler, not a private one."""
import glob
import json
import os
import time

import numpy as np
import torch

from .common import HBM_PEAK_GBS, METRIC, ROOT, log, make_emitter

# (round 4: the padding detector is part of the log-mel stage -- wt_logmel_pad_batch: a one-wave-per-window pass behind
#  the finalising one, which starts its walk at the last valid column; rounds 1-3 timed a separate "padding" stage)
STAGES = ["logmel", "cost", "dtw", "logprob"]
# kernels of each stage as rocprofv3 names them (profiles/*traffic.json keys)
STAGE_KERNELS = {"logmel": ["stft_mel_kernel", "logmel_finalize_kernel", "logmel_init_kernel", "padding_after_finalize_kernel"],
                 "cost": ["rowmean_kernel", "colnorm_kernel", "fix00_kernel"],
                 "dtw": ["dtw_kernel"], "logprob": ["logprob_gather_kernel"]}


def committed_traffic(stage, workload):
    """HBM bytes per launch of a stage's kernels from the newest committed PMC summary OF THIS WORKLOAD
    (profiles/*traffic*.json written by tools/pmc_traffic.py --workload: rocprofv3 FETCH_SIZE x2 [gfx950 correction] +
    WRITE_SIZE, separate passes of this same bench command).  PMC counters cannot be read from inside the timed run,
    so this is the committed measurement -- or None when no summary of the same workload exists."""
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "*traffic*.json")), reverse=True):
        try:
            data = json.load(open(path))
        except Exception:
            continue
        if data.get("_workload") != workload:
            continue
        tot = 0
        for kname, v in data.items():
            if isinstance(v, dict) and any(k in kname for k in STAGE_KERNELS[stage]):
                tot += int(v.get("hbm_bytes", 0))
        return (tot or None), os.path.basename(path)
    return None, None


def role_kernel(args):
    """Publishes the single-batch-in-flight line as soon as it exists, then the line with `--pipeline` batches in flight."""
    import workloads as WL                              # tests/workloads.py: the synthetic inputs + the in-leg parity check
    emit = make_emitter(args.out)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    dry = args.dry_run
    if dry:
        dev = torch.device("cpu")
    else:
        torch.cuda.set_device(local_rank)                      # rank r of the node drives GPU r
        dev = torch.device("cuda", local_rank)
    sync = (lambda: None) if dry else torch.cuda.synchronize
    dist = None
    force_dist = os.environ.get("WT_BENCH_FORCE_DIST") == "1"      # exercise the RCCL path with a single rank
    ranks_seen = 1
    if world > 1 or force_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        if dry:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)     # nccl IS RCCL on ROCm
        ranks_seen = dist.get_world_size()
    if args.inject_fault == "kernel":
        os.abort()

    cfg = WL.WORKLOADS[args.workload]
    n = cfg["n_chunks"]
    depth = max(1, args.pipeline)
    if dry:
        w = dict(cfg=cfg, jumps=torch.zeros(n * (cfg["T"] + 1), dtype=torch.int32), logprob=torch.zeros(n * cfg["T"]), descs=[None] * n)
        sets, pipe, single, fused = [w], None, None, False
        schedule, sub_batches = "serial", 1
    else:
        from whisper_timestamped.pipeline import HotPathPipeline, choose_schedule
        w = WL.make_workload(dev, cfg, seed=1234 + rank)
        if args.align != "auto":
            WL.bind_batch(w, args.align)
        fused = w["batch"].fused_small_units
        sets = [w] + [WL.twin(w) for _ in range(depth - 1)]         # `depth` output-buffer sets over the same inputs
        n_units = len(w["descs"])
        schedule = choose_schedule(args.schedule, n_units, fused)
        sub_batches = args.sub_batches
        if sub_batches == 0:                                        # auto
            sub_batches = 1
        if sub_batches > 1 and (fused or cfg.get("units_per_chunk")):
            sub_batches = 1
        if sub_batches > 1 and args.schedule == "auto":
            schedule = "hilo"                                       # chunk ranges of <= 128 units each: the rule's own case
        # the single-stream pass (stage times, roofline): everything on the current stream, in order
        single = HotPathPipeline(dev, depth=1, schedule="serial", timeline=True)
        pipe = HotPathPipeline(dev, depth=depth, schedule=schedule, sub_batches=sub_batches, rows_per_chunk=cfg.get("T")) \
            if depth > 1 else single
    cfg = w["cfg"]

    gatherers = None
    if world > 1 or force_dist:
        from whisper_timestamped.sharding import ResultGatherer
        gatherers = [ResultGatherer(dist, s_["jumps"].numel(), s_["logprob"].numel(), dev, every=args.gather_every)
                     for s_ in (sets if not dry else [w] * depth)]

    rank_seconds = []            # N > 1: per timed region, every rank's own seconds (before the closing barrier)
    use_gather = [True]          # (switched off for the "what does the gather cost" regions at the end)

    def full_step(k=0, pipelined=False):
        j = k % depth if pipelined else 0
        if dry:
            time.sleep(2e-4)
            if gatherers is not None and use_gather[0]:
                gatherers[j].gather(w["jumps"], w["logprob"])
            return
        last = (pipe if pipelined else single).submit(sets[j]["batch"])
        if gatherers is not None and use_gather[0]:
            with torch.cuda.stream(last):
                gatherers[j].gather(sets[j]["jumps"], sets[j]["logprob"])

    def drain(which):
        if gatherers is not None:
            for g_ in which:
                g_.drain()

    for k in range(args.warmup):
        full_step(k)
    sync()
    if depth > 1:
        for k in range(max(args.warmup, 2 * depth)):              # every stream's scratch arenas exist before the timing
            full_step(k, pipelined=True)
        drain(gatherers or [])
        sync()

    stage_samples = {s: [] for s in STAGES}

    def timed_region(pipelined=False):
        """EXACTLY args.steps steps between barrier + synchronize on both sides; max over ranks; seconds."""
        if single is not None:
            single.sets[0].start_timeline(True)           # (the previous region's stage times have been read: events recycled)
        if dist is not None:
            dist.barrier()
        sync()
        t0 = time.perf_counter()
        for k in range(args.steps):
            full_step(k, pipelined)
        drain((gatherers if pipelined else gatherers[:1]) if gatherers is not None else [])
        sync()
        t_done = time.perf_counter()
        if dist is not None:
            dist.barrier()
        el = time.perf_counter() - t0
        if dist is not None:
            # every rank's own region time (its last kernel / last gather done -> before the closing barrier), then MAX
            mine = torch.tensor([t_done - t0], dtype=torch.float64, device=dev)
            every = [torch.zeros_like(mine) for _ in range(dist.get_world_size())]
            dist.all_gather(every, mine)
            rank_seconds.append([float(x.item()) for x in every])
            te = torch.tensor([el], dtype=torch.float64, device=dev)
            dist.all_reduce(te, op=dist.ReduceOp.MAX)
            el = float(te.item())
        if dry:
            for s in STAGES:
                stage_samples[s].extend([0.04] * args.steps)
        elif not pipelined or depth == 1:
            for stage, ev0, ev1 in single.sets[0].timeline:      # events on the stream the kernels were launched on
                stage_samples[stage].append(ev0.elapsed_time(ev1))
            if fused:
                stage_samples["dtw"].extend([0.0] * args.steps)
        return el

    def measure(pipelined):
        regions = [timed_region(pipelined)]
        n_regions = args.repeats or int(min(2000, max(5, np.ceil(args.min_seconds / max(regions[0], 1e-6)))))
        while len(regions) < n_regions:              # (every rank derives the same count from the max-reduced first region)
            regions.append(timed_region(pipelined))
        return regions

    def check_results(which):
        """sanity inside the bench: the ridge is recovered and log-probs are finite (every buffer set given)"""
        if dry:
            return
        torch.cuda.synchronize()
        for c in which[1:]:
            assert torch.equal(c["host_result"], w["host_result"]), "pipelined steps disagree with the first buffer set"
        hj = w["host_jumps"].numpy()
        devs = []
        for k, d in enumerate(w["descs"]):
            Tk, Fk, j0 = int(d["T"]), int(d["F"]), int(d["jumps_offset"])
            j = hj[j0:j0 + Tk + 1]
            assert j[0] == 0 and j[-1] == Fk - 1 and (np.diff(j) >= 0).all()
            devs.append(np.abs(j[:-1] - np.asarray(w["stairs"][k])))
        assert np.median(np.concatenate(devs)) <= 3
        assert np.isfinite(w["host_logprob"].numpy()).all()

    extras = {}                  # parity_in_leg, per_rank, result_gather_share: filled in as they are measured

    def line(regions, single_regions, batches_in_flight):
        elapsed = float(np.median(regions))
        stage_ms = {s: float(np.median(stage_samples[s])) for s in STAGES}
        ab = WL.algorithmic_bytes(cfg, fused)
        dom = max(stage_ms, key=stage_ms.get)
        achieved = ab[dom] / (stage_ms[dom] * 1e-3) / 1e9
        stages = {s: {"ms": round(stage_ms[s], 4), "alg_MB": round(ab[s] / 1e6, 2),
                      "GBps": round(ab[s] / (stage_ms[s] * 1e-3) / 1e9, 1) if stage_ms[s] else None,
                      "frac_hbm": round(ab[s] / (stage_ms[s] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if stage_ms[s] else None} for s in STAGES}
        traffic, traffic_src = committed_traffic(dom, args.workload)
        ms_per_step = elapsed / args.steps * 1e3
        step_bytes = sum(ab.values())
        step_gbps = step_bytes / (ms_per_step * 1e-3) / 1e9
        in_flight = pipe is not None and batches_in_flight > 1
        sched = pipe.describe() if in_flight else (single.describe() if single is not None else {"schedule": "serial"})
        return {
            "metric": METRIC,
            "value": round(world * n * 30.0 * args.steps / elapsed, 1),
            "unit": "audio-seconds/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
            "data": "synthetic" if not dry else "DRY RUN: no kernels ran, the numbers mean nothing",
            "config": {"workload": cfg["desc"], "units_per_step_per_gpu": n, "units_per_step": len(w["descs"]) if not dry else n,
                       "padded_units": f"1 chunk in {WL.PADDED_EVERY} ends in silence (pad_from = U[F/2, F), its PCM zero from there): the "
                                       "padding detector and the pad mask run inside the timed region",
                       "stages": STAGES,
                       "arithmetic": "f32 cost / log-softmax / log-mel (as the reference's torch CPU ops), f64 DTW (as dtw-python)",
                       "dtw_oracle": "published dtw-python algorithm (symmetric1, strict-< tie order), unpinned against the "
                                     "package itself: absent from the image (tests/test_oracle.py pins oracle/dtw_ref.c AND the HIP "
                                     "kernel on exhaustive path enumeration, on a generic step-pattern interpreter written from "
                                     "the pattern rows the reference builds, and on transformers' DTW for tie-free inputs)",
                       "batches_in_flight": batches_in_flight,
                       "schedule": sched.get("schedule", "serial"), "schedule_streams": sched.get("streams"),
                       "sub_batches": sched.get("sub_batches", 1),
                       "alignment_entry": ("whisper_timestamped.pipeline.HotPathPipeline.submit -> " if not dry else "") +
                                          ("wt_align_batch_v3 (batched row pass + fused small-unit tail kernel; timed as the cost stage)"
                                           if fused else "wt_cost_batch + wt_dtw_batch"),
                       "rccl_ranks_seen": ranks_seen, "cpu_threads_per_rank": int(torch.get_num_threads()),
                       "result_gather": f"{'gloo (dry run)' if dry else 'rccl'} gather to rank 0, one message per {args.gather_every} steps"
                                        if gatherers is not None else "none"},
            "timing": {"regions": len(regions), "steps_per_region": args.steps, "statistic": "median region",
                       "ms_per_step_min": round(min(regions) / args.steps * 1e3, 4),
                       "ms_per_step_max": round(max(regions) / args.steps * 1e3, 4),
                       "timed_seconds_total": round(float(sum(regions)), 3)},
            "single_batch_in_flight": {"ms_per_step": round(float(np.median(single_regions)) / args.steps * 1e3, 4),
                                       "value": round(world * n * 30.0 * args.steps / float(np.median(single_regions)), 1),
                                       "regions": len(single_regions),
                                       "note": "one stream, stages back to back: the run the stage times and the roofline below are from"},
            "roofline": {"bound": "hbm", "kernel": dom, "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_unit": "bytes/launch",
                         "traffic_source": traffic_src, "algorithmic_bytes": ab[dom],
                         "achievable_copy_GBps_guide": 6290.0, "achievable_read_GBps_probe": 6600.0},
            # every algorithmic byte of the step (all four stages) over the step's time: the whole path against the same peak
            "whole_step": {"algorithmic_MB": round(step_bytes / 1e6, 2), "GBps": round(step_gbps, 1),
                           "frac_hbm": round(step_gbps / HBM_PEAK_GBS, 4),
                           "hbm_bound_stages_solo_ms": round(sum(stage_ms[s] for s in ("cost", "logprob")) + 0.0, 4)},
            "stages": stages,
            **extras,
        }

    single_regions = measure(False)                  # one batch in flight: also the per-stage times and the roofline
    check_results(sets[:1])
    if not dry:
        # a few units of what the timed region has just computed, against the oracle (every rank checks its own batch)
        extras["parity_in_leg"] = WL.parity_in_leg(w)
        assert extras["parity_in_leg"]["ok"], extras["parity_in_leg"]
    if rank == 0:
        emit(line(single_regions, single_regions, 1))     # published before the multi-stream pass starts
    mark = len(rank_seconds)
    if depth > 1 and not dry:
        # the parity check above kept the GPU idle for seconds (the oracle runs on the host): the --warmup steps again, on the
        # pipelined path, before its regions are timed
        for k in range(max(args.warmup, 2 * depth)):
            full_step(k, pipelined=True)
        drain(gatherers or [])
        sync()
    regions = measure(True) if depth > 1 else single_regions
    check_results(sets)
    if not dry and depth > 1:
        # the pipelined path's results ARE the single-stream pass's (same inputs, the oracle-checked record above)
        extras["pipelined_equals_single_stream"] = True     # (check_results asserted it buffer set by buffer set)
    if dist is not None and world > 1:
        # N > 1: what every rank needed for the same region (a bad scaling curve can be read: one slow GPU, or all of
        # them waiting), and what the result gather to rank 0 costs (the same regions once more without it)
        per = np.median(np.asarray(rank_seconds[mark:] if depth > 1 else rank_seconds), axis=0) / args.steps * 1e3
        extras["per_rank"] = {"ms_per_step": [round(float(x), 4) for x in per], "min": round(float(per.min()), 4),
                              "max": round(float(per.max()), 4), "skew_max_over_min": round(float(per.max() / per.min()), 4),
                              "note": "each rank's own time from the opening barrier to its last kernel / gather done, "
                                      "median over the timed regions; the headline is the max over ranks incl. the closing barrier"}
        if gatherers is not None:
            use_gather[0] = False
            bare = [timed_region(depth > 1) for _ in range(5)]
            use_gather[0] = True
            extras["result_gather"] = {"ms_per_step_without_gather": round(float(np.median(bare)) / args.steps * 1e3, 4),
                                       "share_of_step": round(max(0.0, 1.0 - float(np.median(bare)) / float(np.median(regions))), 4)}
    if rank == 0:
        emit(line(regions, single_regions, depth))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
