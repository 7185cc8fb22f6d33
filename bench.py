#!/usr/bin/env python3
"""bench.py -- whisper-timestamped alignment hot path on MI355X.

One "step" = one pass of the hot path over one batch of synthetic input that
is already resident in HBM:  log-mel STFT front end -> padding detector ->
local-cost construction (head select + median + softmax + head mean + column
norm) -> DTW + backtrack + jumps -> chosen-token log-softmax gather
(confidence), then an async copy of the (KB-sized) jumps/log-probs to the host.

Workload at N=1 (BASELINE.json configs[1]): whisper-base, a batch of 32
synthetic 30 s chunks; every chunk is one full-window alignment unit
(A=8 alignment heads, T=224 tokens, F=1500 frames: the reference's
trust_whisper_timestamps=False shape, SURVEY.md 8(d) "K-full"), V=51865.
For N>1 every rank owns its own 32 chunks (units are independent: weak
scaling, no data-path collective); the per-step result records are gathered
to rank 0 over RCCL, which is where the reference assembles words.

Two batches are in flight by default (--pipeline 2: step k runs on HIP stream
k % 2 with its own output buffers, no cross-stream dependency): the DTW of a
32-unit batch occupies 32 of the 256 CUs for a fifth of a step and is a
latency chain, so the other batch's HBM-bound kernels run beside it.  The
single-batch-in-flight time is reported in the same line, and the per-stage
times / roofline are measured in that single-stream pass.

Timing: the region of EXACTLY --steps steps (barrier + synchronize on both sides,
max over ranks) is repeated until at least --min-seconds of GPU work have been
timed (never fewer than 5 regions); ms_per_step / value are the MEDIAN region,
min and max are reported next to it.

The same line carries the transcribe()-level number ("e2e"): 32 synthetic 30 s
chunks per sub-batch through log-mel -> whisper-base encoder -> teacher-forced
decoder (fixed synthetic transcript) -> wt_qk_rows_batch -> ONE wt_align_batch
-> ONE wt_logprob_gather_rows -> words (whisper_timestamped/batched.py), with
the share of the alignment kernels in the GPU time and the same chunks through
the reference-shaped CPU path (oracle/, same model on the CPU) beside it.

Process structure: the process the driver starts is an orchestrator that never touches the GPU.  The kernel-level
measurement, the CPU baselines and each transcribe()-level leg run in CHILD processes of this script which publish
their results as they go (the single-stream line before the multi-stream pass starts, the fp32 e2e leg before the
half-precision ones ...): a GPU fault in any leg costs that leg -- reported as {"error": "signal 6"} -- not the line.
`python bench.py --gpus N` outside a launcher re-executes itself under torch.distributed.run (one rank per GPU).

Prints ONE JSON line (rank 0).  metric = audio-seconds aligned per second.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "whisper-timestamped_amd"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8 TB/s spec (6.29 TB/s measured float4 copy; tools/probes/read_probe: 6.6 read-only)

WORKLOADS = {
    # name: (n_chunks, units per chunk generator)
    "kfull": dict(n_chunks=32, A=8, T=224, F=1500, V=51865, n_mels=80,
                  desc="whisper-base, 32 x 30 s chunks, one (8 heads,224 tokens,1500 frames) unit per chunk, V=51865"),
    # secondary workloads (not the BASELINE metric config; DESIGN.md section 6 quotes them)
    "kfull256": dict(n_chunks=256, A=8, T=224, F=1500, V=51865, n_mels=80,
                     desc="whisper-base shapes, 256 x 30 s chunks (one DTW unit per CU), V=51865"),
    "kreal": dict(n_chunks=32, A=8, T=None, F=None, V=51865, n_mels=80, units_per_chunk=5,
                  desc="whisper-base, 32 x 30 s chunks, 5 segments per chunk with the (T, F) mix measured on the reference "
                       "goldens (T p50 11 / p90 30, F p50 144 / p90 352), V=51865"),
    "largev3_fp16": dict(n_chunks=256, A=10, T=224, F=1500, V=51866, n_mels=128, qk_dtype="f16",
                         desc="whisper-large-v3 shapes (BASELINE config 5): 256 x 30 s chunks, 10 heads, fp16 QK rows, "
                              "128 mels, V=51866"),
}


def result_buffers(n_jumps, n_logprob, dev):
    """The KB-sized results of a step live in ONE device record (jumps, then the log-probs) so that a single async
    copy brings them to the host (and a single message carries them to rank 0)."""
    rec = torch.empty(n_jumps + n_logprob, dtype=torch.int32, device=dev)
    host = torch.empty(n_jumps + n_logprob, dtype=torch.int32).pin_memory()
    return dict(result=rec, jumps=rec[:n_jumps], logprob=rec[n_jumps:].view(torch.float32), host_result=host,
                host_jumps=host[:n_jumps], host_logprob=host[n_jumps:].view(torch.float32))


PADDED_EVERY = 16          # SURVEY.md 8(d) set K: "pad_from = -1 except 1/16 of segments with pad_from = U[F/2, F)"


def padded_chunks(n, F, rs):
    """Which chunks end in silence, and from which 20 ms frame on: chunk b (b % 16 == 7) holds only pad_from * 320 real
    samples -- its log-mel is exact zeros from column 2 * pad_from (pad_or_trim), find_start_padding returns that
    column, and T.py:1554-1565 masks the cost matrix from frame pad_from = column // 2."""
    pad = np.full(n, -1, dtype=np.int64)
    for b in range(n):
        if b % PADDED_EVERY == PADDED_EVERY // 2 - 1:
            pad[b] = int(rs.randint(F // 2, F))
    return pad


def make_workload(dev, cfg, seed):
    from whisper_timestamped import _lib
    if cfg.get("units_per_chunk"):
        return make_workload_kreal(dev, cfg, seed)
    n, A, T, F, V = cfg["n_chunks"], cfg["A"], cfg["T"], cfg["F"], cfg["V"]
    g = torch.Generator(device=dev).manual_seed(seed)
    qk = torch.randn((n, A, T, 1500), generator=g, device=dev, dtype=torch.float32)
    qk_half = cfg.get("qk_dtype") == "f16"
    # monotone ridge (+6 on a token->frame staircase, 3 frames wide): SURVEY.md 8(d) set K
    rs = np.random.RandomState(seed)
    pad = padded_chunks(n, F, rs)
    stairs = np.stack([np.sort(rs.randint(0, F if pad[b] < 0 else pad[b], size=T)) for b in range(n)])
    st = torch.from_numpy(stairs).to(dev)
    fr = torch.arange(1500, device=dev).view(1, 1, 1500)
    ridge = ((fr - st.unsqueeze(-1)).abs() <= 1).to(torch.float32) * 6.0
    qk += ridge.unsqueeze(1)
    del ridge
    sample = parity_sample_units(pad)
    qk_f32_sample = {b: qk[b].cpu() for b in sample} if qk_half else None     # (what the fp16 rows were rounded from)
    if qk_half:
        qk = qk.half()
    logits = torch.randn((n * T, V), generator=g, device=dev, dtype=torch.float32) * 3.0
    tokens = torch.randint(0, V, (n * T,), generator=g, device=dev, dtype=torch.int32)
    pcm = torch.randn((n, 480000), generator=g, device=dev, dtype=torch.float32) * 0.1
    n_valid = np.where(pad >= 0, pad * 320, 480000).astype(np.int32)
    for b in np.nonzero(pad >= 0)[0]:
        pcm[b, int(n_valid[b]):] = 0.0
    from whisper_timestamped.audio import mel_filters
    fb = mel_filters(dev, cfg["n_mels"])
    descs = _lib.make_descs(n)
    for b, d in enumerate(descs):
        d["qk_offset"], d["head_stride"], d["row_stride"] = b * A * T * 1500, T * 1500, 1500
        d["T"], d["F"], d["start_token"], d["pad_from"] = T, F, 0, int(pad[b])
    n_cost, n_jumps, n_path = _lib.layout_outputs(descs)
    cfg = dict(cfg, pad_from=[int(x) for x in pad], n_valid=[int(x) for x in n_valid])
    w = dict(cfg=cfg, qk=qk, logits=logits, tokens=tokens, pcm=pcm, fb=fb, descs=descs,
             n_valid=torch.from_numpy(n_valid).to(dev), parity_units=sample, qk_f32_sample=qk_f32_sample,
             unit_chunk=list(range(n)), unit_row0=[0] * n, unit_logit_row0=[b * T for b in range(n)],
             descs_dev=_lib.descs_to_device(descs, dev), head_idx=torch.arange(A, dtype=torch.int32, device=dev),
             cost=torch.empty(n_cost, dtype=torch.float32, device=dev),
             **result_buffers(n_jumps, n * T, dev),
             mel=torch.empty((n, cfg["n_mels"], 3000), dtype=torch.float32, device=dev),
             gmax=torch.empty(n, dtype=torch.float32, device=dev),
             pad=torch.empty(n, dtype=torch.int32, device=dev),
             stairs=stairs)
    return w


def parity_sample_units(pad):
    """The units the in-leg parity check compares with the oracle: the first one, and the first padded one."""
    padded = [int(b) for b in np.nonzero(pad >= 0)[0][:1]]
    return sorted(set([0] + padded))


def parity_in_leg(w):
    """A few units of the batch the timed region has just processed, through the oracle (oracle/: the CPU restatement of
    the reference; used here as the CHECKER, never as the thing measured): the jumps of the last timed step must be the
    oracle's for the same logits -- bit for bit with fp32 rows; with fp16 rows (a storage option the reference does not
    have) against the oracle on the same rounded logits AND on the fp32 logits they were rounded from (max |d frame|) --
    the log-probabilities within 2e-5, the padding index exact, the log-mel within 2e-4."""
    from oracle import align_ref as O
    cfg = w["cfg"]
    A, V = cfg["A"], cfg["V"]
    torch.cuda.synchronize()
    hj, hl = w["host_jumps"].numpy(), w["host_logprob"].numpy()
    pad_dev = w["pad"].cpu().numpy()
    out = {"units": [], "jumps_equal_oracle": True, "max_abs_dlogprob": 0.0, "padding_index_equal_oracle": True}
    qk = w["qk"]
    rows = qk.shape[2]
    worst_half = 0
    for k in w["parity_units"]:
        d = w["descs"][k]
        T, F, start, pf = int(d["T"]), int(d["F"]), int(d["start_token"]), int(d["pad_from"])
        b, r0 = w["unit_chunk"][k], w["unit_row0"][k]
        sel = qk[b, :, r0:r0 + T, start:start + F].float().cpu()
        cost = O.cost_matrix_ref(sel, 9, 1.0, pf if pf > 0 else None, start)
        r = O.dtw_ref(cost)
        want = O.jumps_from_path(r.index1s, r.index2s)
        j0 = int(d["jumps_offset"])
        got = hj[j0:j0 + T + 1]
        same = bool(np.array_equal(got, want))
        out["jumps_equal_oracle"] &= same
        rec = {"unit": int(k), "T": T, "F": F, "pad_from": pf, "jumps_equal": same}
        if w.get("qk_f32_sample") is not None:
            sel32 = w["qk_f32_sample"][b][:, r0:r0 + T, start:start + F]
            r32 = O.dtw_ref(O.cost_matrix_ref(sel32, 9, 1.0, pf if pf > 0 else None, start))
            df = int(np.abs(O.jumps_from_path(r32.index1s, r32.index2s) - got).max())
            rec["max_dframe_vs_fp32_oracle_on_the_fp32_logits"] = df
            worst_half = max(worst_half, df)
        l0 = w["unit_logit_row0"][k]
        ref = O.token_logprob_gather_ref(w["logits"][l0:l0 + T].cpu(), w["tokens"][l0:l0 + T].cpu().numpy()).numpy()
        out["max_abs_dlogprob"] = max(out["max_abs_dlogprob"], float(np.abs(ref - hl[l0:l0 + T]).max()))
        nv = int(cfg["n_valid"][b])
        mel_ref = O.pad_or_trim_ref(O.log_mel_spectrogram_ref(w["pcm"][b, :nv].cpu(), cfg["n_mels"]), 3000)
        rec["max_abs_dlogmel"] = float((w["mel"][b].cpu() - mel_ref).abs().max())
        sp = O.find_start_padding_ref(mel_ref[None])
        out["padding_index_equal_oracle"] &= (int(pad_dev[b]) == (-1 if sp is None else int(sp)))
        out["units"].append(rec)
    out["max_abs_dlogprob"] = float(f"{out['max_abs_dlogprob']:.3g}")
    out["max_abs_dlogmel"] = float(f"{max(u['max_abs_dlogmel'] for u in out['units']):.3g}")
    if w.get("qk_f32_sample") is not None:
        out["fp16_rows_max_dframe_vs_fp32_oracle"] = worst_half
    ok = out["jumps_equal_oracle"] and out["padding_index_equal_oracle"] and out["max_abs_dlogprob"] <= 2e-5 and \
        out["max_abs_dlogmel"] <= 2e-4
    out["ok"] = bool(ok)
    return out


def make_workload_kreal(dev, cfg, seed):
    """Several short units per chunk: each unit is a window [start, start+F) of T consecutive rows of the chunk's
    captured QK block (the layout the capture ring produces)."""
    from whisper_timestamped import _lib
    from whisper_timestamped.audio import mel_filters
    import synth
    n, A, V, U = cfg["n_chunks"], cfg["A"], cfg["V"], cfg["units_per_chunk"]
    g = torch.Generator(device=dev).manual_seed(seed)
    Ts, Fs = synth.draw_real_shapes(seed, n * U)
    rows_per_chunk = 256
    qk = torch.randn((n, A, rows_per_chunk, 1500), generator=g, device=dev, dtype=torch.float32)
    rs = np.random.RandomState(seed)
    pad = padded_chunks(n, 1500, rs)           # the chunk's mel is zero from column 2 * pad[b]: max_duration = pad[b]
    raw, tot_T = [], 0
    for b in range(n):
        row = 0
        for u in range(U):
            k = b * U + u
            T, F = int(min(Ts[k], rows_per_chunk - row - 1)), int(Fs[k])
            T = max(T, 2)
            F = max(F, T + 1)
            start = int(rs.randint(0, 1500 - F + 1))
            # T.py:1561-1565: the mask applies when the window starts before max_duration, and is then applied at the
            # ABSOLUTE index used as a relative column (the reference's quirk): columns >= pad[b] of the window
            pf = int(pad[b]) if (pad[b] >= 0 and start < pad[b]) else -1
            st = np.sort(rs.randint(0, F if (pf < 0 or pf >= F) else max(pf, 1), size=T))
            for t in range(T):
                a, e = start + max(st[t] - 1, 0), start + min(st[t] + 2, F)
                qk[b, :, row + t, a:e] += 6.0
            raw.append(dict(qk_offset=(b * A * rows_per_chunk + row) * 1500, T=T, F=F, start=start, stairs=st, pad_from=pf,
                            chunk=b, row0=row, logit_row0=tot_T))
            row += T
            tot_T += T
    order = _lib.launch_order([(r["T"], r["F"]) for r in raw])      # grouped by F class, as AlignmentBatch does
    descs = _lib.make_descs(n * U)
    stairs = []
    for d, i in zip(descs, order):
        r = raw[i]
        d["qk_offset"], d["head_stride"], d["row_stride"] = r["qk_offset"], rows_per_chunk * 1500, 1500
        d["T"], d["F"], d["start_token"], d["pad_from"] = r["T"], r["F"], r["start"], r["pad_from"]
        stairs.append(r["stairs"])
    n_valid = np.where(pad >= 0, pad * 320, 480000).astype(np.int32)
    masked = [k for k, i in enumerate(order) if 0 < raw[i]["pad_from"] < raw[i]["F"]]
    sample = sorted(set([0, len(order) // 2] + masked[:1]))
    n_cost, n_jumps, n_path = _lib.layout_outputs(descs)
    logits = torch.randn((tot_T, V), generator=g, device=dev, dtype=torch.float32) * 3.0
    tokens = torch.randint(0, V, (tot_T,), generator=g, device=dev, dtype=torch.int32)
    pcm = torch.randn((n, 480000), generator=g, device=dev, dtype=torch.float32) * 0.1
    for b in np.nonzero(pad >= 0)[0]:
        pcm[b, int(n_valid[b]):] = 0.0
    cfg = dict(cfg, n_rows=tot_T, units=[(int(d["T"]), int(d["F"])) for d in descs], pad_from=[int(x) for x in pad],
               n_valid=[int(x) for x in n_valid])
    return dict(cfg=cfg, qk=qk, logits=logits, tokens=tokens, pcm=pcm, fb=mel_filters(dev, cfg["n_mels"]), descs=descs,
                n_valid=torch.from_numpy(n_valid).to(dev), parity_units=sample, qk_f32_sample=None,
                unit_chunk=[raw[i]["chunk"] for i in order], unit_row0=[raw[i]["row0"] for i in order],
                unit_logit_row0=[raw[i]["logit_row0"] for i in order],
                descs_dev=_lib.descs_to_device(descs, dev), head_idx=torch.arange(A, dtype=torch.int32, device=dev),
                cost=torch.empty(n_cost, dtype=torch.float32, device=dev),
                **result_buffers(n_jumps, tot_T, dev),
                mel=torch.empty((n, cfg["n_mels"], 3000), dtype=torch.float32, device=dev),
                gmax=torch.empty(n, dtype=torch.float32, device=dev), pad=torch.empty(n, dtype=torch.int32, device=dev),
                stairs=stairs)


# (round 4: the padding detector is part of the log-mel stage -- wt_logmel_pad_batch: a one-wave-per-window pass behind
#  the finalising one, which starts its walk at the last valid column; rounds 1-3 timed a separate "padding" stage)
STAGES = ["logmel", "cost", "dtw", "logprob"]
# kernels of each stage as rocprofv3 names them (profiles/*traffic.json keys)
STAGE_KERNELS = {"logmel": ["stft_mel_kernel", "logmel_finalize_kernel", "logmel_init_kernel", "padding_after_finalize_kernel"], "cost": ["rowmean_kernel", "colnorm_kernel", "fix00_kernel"],
                 "dtw": ["dtw_kernel"], "logprob": ["logprob_gather_kernel"]}


def committed_traffic(stage, workload):
    """HBM bytes per launch of a stage's kernels from the newest committed PMC summary OF THIS WORKLOAD
    (profiles/*traffic*.json written by tools/pmc_traffic.py --workload: rocprofv3 FETCH_SIZE x2 [gfx950 correction] +
    WRITE_SIZE, separate passes of this same bench command).  PMC counters cannot be read from inside the timed run,
    so this is the committed measurement -- or None when no summary of the same workload exists."""
    import glob
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


def _stage_calls(w):
    from whisper_timestamped import _lib
    L = _lib.load()
    cfg = w["cfg"]
    n, V = cfg["n_chunks"], cfg["V"]
    n_units = len(w["descs"])
    n_rows = cfg.get("n_rows") or n * cfg["T"]

    def logmel(st):
        _lib._check(L.wt_logmel_pad_batch(w["pcm"].data_ptr(), n, 480000, w["n_valid"].data_ptr(), w["fb"].data_ptr(), cfg["n_mels"],
                                          3000, w["mel"].data_ptr(), w["gmax"].data_ptr(), w["pad"].data_ptr(), st),
                    "wt_logmel_pad_batch")

    def cost(st):
        _lib._check(L.wt_cost_batch(w["qk"].data_ptr(), 1 if cfg.get("qk_dtype") == "f16" else 0, w["descs"].ctypes.data, w["descs_dev"].data_ptr(), n_units,
                                    w["head_idx"].data_ptr(), cfg["A"], 9, 1.0, w["cost"].data_ptr(), st), "wt_cost_batch")

    def dtw(st):
        _lib._check(L.wt_dtw_batch(w["cost"].data_ptr(), w["descs"].ctypes.data, w["descs_dev"].data_ptr(), n_units,
                                   w["jumps"].data_ptr(), 0, 0, 0, 0, st), "wt_dtw_batch")

    if w.get("align") == "fused":
        # ONE entry point for cost + DTW (wt_align_batch_v3): after the batched row pass, units of the per-segment shape
        # take the fused tail kernel (column norm, cost[0,0], DTW, backtrack in one workgroup, the matrix in LDS), the
        # others the batched kernels.  The whole of it is timed as the "cost" stage; the "dtw" stage is empty.
        def cost(st):   # noqa: F811
            _lib._check(L.wt_align_batch_v3(w["qk"].data_ptr(), 1 if cfg.get("qk_dtype") == "f16" else 0, w["descs"].ctypes.data,
                                            w["descs_dev"].data_ptr(), n_units, w["head_idx"].data_ptr(), cfg["A"], 9, 1.0,
                                            w["cost"].data_ptr(), w["jumps"].data_ptr(), 0, 0, 0, 0, 0, st), "wt_align_batch_v3")

        def dtw(st):    # noqa: F811
            pass

    def logprob(st):
        _lib._check(L.wt_logprob_gather_batch(w["logits"].data_ptr(), 0, V, n_rows, V, w["tokens"].data_ptr(), 0, 0,
                                              w["logprob"].data_ptr(), st), "wt_logprob_gather_batch")

    return dict(logmel=logmel, cost=cost, dtw=dtw, logprob=logprob)


# stage -> lane: with --overlap the three lanes run on three HIP streams (the stages of one lane stay ordered)
LANES = [["logmel"], ["cost", "dtw"], ["logprob"]]


def cu_masked_streams(dev, n_dtw_cus):
    """Two HIP streams with complementary CU masks (hipExtStreamCreateWithCUMask).  Mask bit i is CU i in the
    driver's enumeration, which interleaves the XCDs (bit i -> XCD i % 8): the first n bits are n/8 CUs on every XCD."""
    import ctypes
    hip = ctypes.CDLL("libamdhip64.so")
    n_cus = torch.cuda.get_device_properties(dev).multi_processor_count
    words = (n_cus + 31) // 32

    def make(bits):
        mask = (ctypes.c_uint32 * words)()
        for b in bits:
            mask[b // 32] |= 1 << (b % 32)
        st = ctypes.c_void_p()
        rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(st), ctypes.c_uint32(words), mask)
        assert rc == 0, f"hipExtStreamCreateWithCUMask failed: {rc}"
        return torch.cuda.ExternalStream(st.value, device=dev)
    return make(range(n_dtw_cus)), make(range(n_dtw_cus, n_cus))


def run_step(w, ev=None, streams=None):
    """One pass of the hot path.  ev: optional {stage: (start_event, end_event)}, recorded on the stream the stage's
    kernels are launched on.  streams: None = everything on the current stream, in order; else 3 torch streams."""
    calls = w.setdefault("_calls", _stage_calls(w))
    main = torch.cuda.current_stream()
    if isinstance(streams, tuple) and streams[0] == "dtw_logmel":
        # cost -> [DTW on a side stream: 32 CUs, 141 KB of LDS each, latency-bound] || [log-mel: VALU-bound, fills
        # the other 224 CUs; its workgroups do not fit next to a DTW workgroup] -> log-prob alone (HBM-bound)
        side = streams[1]
        if ev: ev["cost"][0].record(main)
        calls["cost"](main.cuda_stream)
        if ev: ev["cost"][1].record(main)
        side.wait_stream(main)
        if ev: ev["dtw"][0].record(side)
        calls["dtw"](side.cuda_stream)
        if ev: ev["dtw"][1].record(side)
        for stage in ("logmel",):
            if ev: ev[stage][0].record(main)
            calls[stage](main.cuda_stream)
            if ev: ev[stage][1].record(main)
        main.wait_stream(side)
        if ev: ev["logprob"][0].record(main)
        calls["logprob"](main.cuda_stream)
        if ev: ev["logprob"][1].record(main)
    elif isinstance(streams, tuple) and streams[0] == "cumask":
        # log-mel, padding, cost on the whole chip; then the DTW on its own CUs (one workgroup = one CU per unit) beside
        # the HBM-bound log-prob gather on the other CUs: CU-masked streams, so neither kernel's waves land on the
        # other's CUs (without masks the gather's waves share the DTW's SIMDs and both kernels slow down)
        dtw_s, lp_s = streams[1], streams[2]
        for stage in ("logmel", "cost"):
            if ev: ev[stage][0].record(main)
            calls[stage](main.cuda_stream)
            if ev: ev[stage][1].record(main)
        dtw_s.wait_stream(main)
        lp_s.wait_stream(main)
        if ev: ev["dtw"][0].record(dtw_s)
        calls["dtw"](dtw_s.cuda_stream)
        if ev: ev["dtw"][1].record(dtw_s)
        if ev: ev["logprob"][0].record(lp_s)
        calls["logprob"](lp_s.cuda_stream)
        if ev: ev["logprob"][1].record(lp_s)
        main.wait_stream(dtw_s)
        main.wait_stream(lp_s)
    elif isinstance(streams, tuple) and streams[0] == "dtw":
        side = streams[1]
        for stage in ("logmel", "cost"):
            if ev: ev[stage][0].record(main)
            calls[stage](main.cuda_stream)
            if ev: ev[stage][1].record(main)
        side.wait_stream(main)                      # the cost matrix is ready
        if ev: ev["dtw"][0].record(side)
        calls["dtw"](side.cuda_stream)
        if ev: ev["dtw"][1].record(side)
        if ev: ev["logprob"][0].record(main)
        calls["logprob"](main.cuda_stream)
        if ev: ev["logprob"][1].record(main)
        main.wait_stream(side)
    elif streams is None:
        # one stream, stages back to back: the end event of a stage IS the start event of the next one
        # (6 event records per step instead of 10: each record is a ~2-3 us marker in the queue)
        order = [stage for lane in LANES for stage in lane]
        if ev: ev[order[0]][0].record(main)
        for stage in order:
            calls[stage](main.cuda_stream)
            if ev: ev[stage][1].record(main)
    else:
        fork = w.setdefault("_fork", torch.cuda.Event())
        fork.record(main)
        for lane, s in zip(LANES, streams):
            s.wait_event(fork)
            for stage in lane:
                if ev: ev[stage][0].record(s)
                calls[stage](s.cuda_stream)
                if ev: ev[stage][1].record(s)
            main.wait_stream(s)
    w["host_result"].copy_(w["result"], non_blocking=True)


# How the stages of ONE batch share the chip when several batches are in flight (--schedule; tools/overlap_matrix.py timed
# the candidates on the current kernels: profiles/r5*_overlap_matrix.json).  "serial": every batch on one stream, stages in
# order (rounds 2-4).  "hilo": per batch a HIGH-priority HIP stream for the kernels that cannot use the chip's bandwidth
# (stft_mel: VALU/LDS-bound; dtw_kernel: a latency chain on 32 CUs) and a LOW-priority one for the HBM-bound ones (cost,
# log-prob gather): the dispatcher places the high-priority workgroups first, the bandwidth kernels fill what is left --
# each HBM-bound kernel runs beside a compute-bound one of the same batch, and the second batch fills the gaps.
STAGE_ORDER = ["logmel", "cost", "dtw", "logprob"]
SCHEDULES = {
    "serial": None,
    "hilo": {"assign": {"logmel": ("hi", "high"), "dtw": ("hi", "high"), "cost": ("lo", "low"), "logprob": ("lo", "low")},
             "order": STAGE_ORDER},
    "dtw_hi": {"assign": {"logmel": ("lo", "low"), "dtw": ("hi", "high"), "cost": ("lo", "low"), "logprob": ("lo", "low")},
               "order": STAGE_ORDER},
    "two_streams": {"assign": {"logmel": ("hi", "normal"), "dtw": ("hi", "normal"), "cost": ("lo", "normal"), "logprob": ("lo", "normal")},
                    "order": STAGE_ORDER},
    # hilo with other ISSUE orders (the assignment is the same): the cost stage's row pass needs 49 KB of LDS per workgroup,
    # the persistent stft_mel launch fills every CU's LDS when it gets there first
    "hilo_cost_first": {"assign": {"logmel": ("hi", "high"), "dtw": ("hi", "high"), "cost": ("lo", "low"), "logprob": ("lo", "low")},
                        "order": ["cost", "logmel", "dtw", "logprob"]},
    "hilo_logmel_last": {"assign": {"logmel": ("hi", "high"), "dtw": ("hi", "high"), "cost": ("lo", "low"), "logprob": ("lo", "low")},
                         "order": ["cost", "dtw", "logprob", "logmel"]},
    # hilo with ONE low-priority stream for the HBM-bound kernels of ALL batches in flight (a key that starts with
    # "shared" names the same stream in every batch): two bandwidth kernels never compete with each other, each runs at
    # its solo speed with a compute-bound kernel of either batch beside it
    "hilo_one_lo": {"assign": {"logmel": ("hi", "high"), "dtw": ("hi", "high"), "cost": ("shared_lo", "low"), "logprob": ("shared_lo", "low")},
                    "order": STAGE_ORDER},
    "hilo_one_lo_normal": {"assign": {"logmel": ("hi", "high"), "dtw": ("hi", "high"), "cost": ("shared_lo", "normal"), "logprob": ("shared_lo", "normal")},
                           "order": STAGE_ORDER},
}


def stream_priorities():
    """{"low": least, "normal": 0, "high": greatest} of hipDeviceGetStreamPriorityRange (MI355X / ROCm 7.2: 1, 0, -1)."""
    import ctypes
    hip = ctypes.CDLL("libamdhip64.so")
    lo, hi = ctypes.c_int(), ctypes.c_int()
    rc = hip.hipDeviceGetStreamPriorityRange(ctypes.byref(lo), ctypes.byref(hi))
    assert rc == 0, f"hipDeviceGetStreamPriorityRange failed: {rc}"
    return {"low": lo.value, "normal": 0, "high": hi.value}


def priority_stream(dev, priority):
    import ctypes
    hip = ctypes.CDLL("libamdhip64.so")
    st = ctypes.c_void_p()
    rc = hip.hipStreamCreateWithPriority(ctypes.byref(st), ctypes.c_uint(1), ctypes.c_int(priority))     # 1 = hipStreamNonBlocking
    assert rc == 0, f"hipStreamCreateWithPriority failed: {rc}"
    return torch.cuda.ExternalStream(st.value, device=dev)


def plan_streams(dev, plan, shared=None):
    """One HIP stream per stream key of the plan; keys that start with "shared" are taken from (and added to) `shared`,
    the dictionary the caller passes for every batch in flight."""
    prio = stream_priorities()
    out = {}
    for stage in plan["order"]:
        key, level = plan["assign"][stage]
        if key in out:
            continue
        if key.startswith("shared") and shared is not None:
            if key not in shared:
                shared[key] = priority_stream(dev, prio[level])
            out[key] = shared[key]
        else:
            out[key] = priority_stream(dev, prio[level])
    return out


def run_step_plan(w, plan, streams):
    """One pass of the hot path with its stages on the streams of `plan`.  The DTW waits for its cost stage; a buffer set's
    NEXT step waits for what still reads its buffers (the DTW reads the cost matrix the next cost stage overwrites, the
    result copy reads the record the next DTW / log-prob gather overwrite); the result copy waits for every stage.
    Returns the stream the copy was queued on (the step is complete when that stream is)."""
    calls = w.setdefault("_calls", _stage_calls(w))
    ev = w.setdefault("_plan_events", {})
    for stage in plan["order"]:
        st = streams[plan["assign"][stage][0]]
        if stage == "dtw":
            st.wait_event(ev["cost"])
        if stage == "cost" and "dtw" in ev:
            st.wait_event(ev["dtw"])
        if (stage in ("dtw", "logprob") or (stage == "cost" and "dtw" not in plan["order"])) and "copy" in ev:
            st.wait_event(ev["copy"])
        calls[stage](st.cuda_stream)
        if stage not in ev:
            ev[stage] = torch.cuda.Event()
        ev[stage].record(st)
    last = streams[plan["assign"][plan["order"][-1]][0]]
    for stage in plan["order"]:
        if streams[plan["assign"][stage][0]] is not last:
            last.wait_event(ev[stage])
    with torch.cuda.stream(last):
        w["host_result"].copy_(w["result"], non_blocking=True)
    if "copy" not in ev:
        ev["copy"] = torch.cuda.Event()
    ev["copy"].record(last)
    return last


def algorithmic_bytes(cfg, fused=False):
    """Per launch (= per step on one rank), SURVEY.md 8(d)."""
    n, A, V, M = cfg["n_chunks"], cfg["A"], cfg["V"], cfg["n_mels"]
    units = cfg.get("units") or [(cfg["T"], cfg["F"])] * n
    s_in = 2 if cfg.get("qk_dtype") == "f16" else 4
    tf = sum(t * f for t, f in units)
    rows = sum(t for t, _ in units)
    n_valid = cfg.get("n_valid") or [480000] * n
    # the real samples in, the whole (M, 3000) window out, one padding index per window (decided while writing it)
    logmel = sum(v * 4 for v in n_valid) + n * M * 3000 * 4 + n * 4
    if fused:   # the same bytes as the two stages below, moved by one entry point and timed as one stage
        return {"logmel": logmel,
                "cost": A * tf * s_in + 2 * tf * 4 + 4 * (rows + len(units)), "dtw": 0, "logprob": rows * (V * 4 + 8)}
    return {
        "logmel": logmel,
        "cost": A * tf * s_in + tf * 4,                    # selected-head logits once, cost once
        "dtw": tf * 4 + 4 * (rows + len(units)),           # read cost once, write jumps
        "logprob": rows * (V * 4 + 8),                     # read each logit row once
    }


def cpu_baseline(cfg, w, budget_s=12.0, threads=None, distinct=32):
    """The oracle (CPU restatement of the reference path) on a bounded sample of the same workload, host cores of this
    box, rank 0 only.  The sample cycles over `distinct` DIFFERENT chunks (no cache-warm repeats of a few inputs).
    threads=1: the reference's alignment is effectively single-threaded (scipy / dtw-python do not thread)."""
    from oracle import align_ref as O
    T = cfg["T"]
    nd = min(distinct, cfg["n_chunks"])
    def to_host(t):
        """device -> page-locked host memory, chunk by chunk (the runtime never has to lock GBs of pageable memory
        on the fly for one copy)"""
        host = torch.empty(t.shape, dtype=t.dtype, pin_memory=True)
        for a in range(0, t.shape[0], 4):
            host[a:a + 4].copy_(t[a:a + 4])
        return host
    qk = to_host(w["qk"][:nd].float())
    logits = to_host(w["logits"][: nd * T].view(nd, T, -1)).view(nd * T, -1)
    tokens = w["tokens"][: nd * T].cpu().numpy()
    pcm = to_host(w["pcm"][:nd])
    before = torch.get_num_threads()
    if threads:
        torch.set_num_threads(threads)
    try:
        done, t0 = 0, time.perf_counter()
        while True:
            b = done % nd
            mel = O.pad_or_trim_ref(O.log_mel_spectrogram_ref(pcm[b][:cfg["n_valid"][b]], cfg["n_mels"]), 3000)
            cost = O.cost_matrix_ref(qk[b][:, :, :cfg["F"]], 9, 1.0, O.max_duration_ref(mel[None]), 0)
            r = O.dtw_ref(cost)
            O.jumps_from_path(r.index1s, r.index2s)
            O.token_logprob_gather_ref(logits[b * T:(b + 1) * T], tokens[b * T:(b + 1) * T])
            done += 1
            el = time.perf_counter() - t0
            if el > budget_s or done >= 2 * nd:
                break
        used = int(torch.get_num_threads())
    finally:
        torch.set_num_threads(before)
    return {"value": round(30.0 * done / el, 2), "unit": "audio-seconds/s", "cores": used,
            "kind": "port",
            "sample": f"{done} 30 s K-full chunks ({min(done, nd)} distinct) through oracle/ (scipy median_filter + torch CPU "
                      f"softmax/mean/norm/log_softmax/stft with {used} intra-op thread(s), single-thread C DTW + backtrack), "
                      f"{el:.1f} s wall"}


# --------------------------------------------------------------------------------------------------- transcribe() level
E2E_SEGMENTS = [(0, 280), (300, 560), (580, 900), (920, 1200), (1220, 1480)]    # 5 timestamped segments per window
E2E_TEXT_PER_SEGMENT = 17                                                       # ~86 text tokens per window (SURVEY 8d set M)


def e2e_transcript(tokenizer, seed):
    """The fixed synthetic transcript of one 30 s window, as whisper hands a window's tokens to the naive strategy with
    trust_whisper_timestamps=False: <|s|> text <|e|><|s'|> text <|e'|> ..."""
    rs = np.random.RandomState(seed)
    ts0 = tokenizer.timestamp_begin
    banned = set(getattr(tokenizer, "non_speech_tokens", ())) | {220}
    toks = []
    for s, e in E2E_SEGMENTS:
        text = [int(t) for t in rs.randint(300, 40000, size=E2E_TEXT_PER_SEGMENT)]
        toks += [ts0 + s] + [t if t not in banned else 300 for t in text] + [ts0 + e]
    return toks


def e2e_cpu_reference_window(W, model_cpu, tokenizer, heads, pcm, window_tokens):
    """One window the way the reference's naive loop does it (transcribe.py:1204-1300), on the CPU through oracle/:
    torch.stft log-mel, the model unfused with every hooked layer's QK observed, log_softmax of the whole (T, V)
    block, the oracle's perform_word_alignment, a Python loop of logprobs[:, step, tok] reads."""
    from oracle import align_ref as O
    import torch.nn.functional as F
    ts0 = tokenizer.timestamp_begin
    mel = O.pad_or_trim_ref(O.log_mel_spectrogram_ref(pcm, model_cpu.dims.n_mels), 3000).unsqueeze(0)
    toks = list(window_tokens)
    while toks[0] >= ts0:
        toks = toks[1:]
    while toks[-1] >= ts0:
        toks = toks[:-1]
    sot = tokenizer.sot_sequence
    if len(sot) == 3:
        sot = (sot[0], tokenizer.to_language_token("en"), sot[2])
    toks = [*sot, ts0] + toks
    i_start = len(sot)
    att = [None] * len(model_cpu.decoder.blocks)
    hooks = [blk.cross_attn.register_forward_hook(lambda m, i, o, k=k: att.__setitem__(k, o[-1]))
             for k, blk in enumerate(model_cpu.decoder.blocks)]
    try:
        with torch.no_grad(), W.model.disable_sdpa():
            logprobs = F.log_softmax(model_cpu(mel, torch.tensor(toks, dtype=torch.int32).unsqueeze(0)), dim=-1)
    finally:
        for h in hooks:
            h.remove()
    end_token = ts0 + round(min(480000, pcm.shape[-1]) // 320)
    toks = toks[i_start:] + [end_token]
    att = [w[:, :, i_start - 1:, :] for w in att]
    ws = O.perform_word_alignment_ref(toks, att, tokenizer, use_space=True, alignment_heads=np.asarray(heads), mfcc=mel,
                                      refine_whisper_precision_nframes=25, detect_disfluencies=False)
    for word in ws:
        ids = word["tokens_indices"]
        lp = [logprobs[:, step, tok] for step, tok in zip(range(i_start, i_start + len(ids)), ids)]
        i_start += len(word["tokens"])
        word["confidence_raw"] = torch.cat(lp).mean().exp().item() if lp else 0.0
        word["mean_logprob_raw"] = torch.cat(lp).mean().item() if lp else None
    return ws


def run_e2e(dev, args, leg, emit):
    """audio-seconds transcribed-with-word-timestamps per second at the transcribe() level (SURVEY 8d "End-to-end
    audio-s/s"): whisper-base, 32 synthetic 30 s chunks per launch set, teacher-forced transcript.  One leg per child
    process: "fp32" (the CPU reference's arithmetic; also the CPU e2e baseline and the word parity against it) or
    "fp16" (half-precision activations, eager).  `emit` publishes
    what has been measured so far: a fault later in the leg cannot take it back."""
    import whisper_double as W          # tests/whisper_double: stand-in for openai-whisper (absent from this image)
    W.install()
    from whisper_timestamped.alignment import head_pairs
    from whisper_timestamped.batched import BatchedAligner, WindowJob, align_windows
    from whisper_timestamped.transcribe import get_alignment_heads
    n_per, steps = args.e2e_windows, args.e2e_steps
    name = args.e2e_model
    model = W.build_model(name, seed=0, device=dev)
    if hasattr(model, "alignment_heads"):
        del model.alignment_heads                          # -> the published whisper-base heads (parameter-count table)
    heads = head_pairs(get_alignment_heads(model))
    tokenizer = W.tokenizer.get_tokenizer(True, language="en", task="transcribe",
                                          **({"num_languages": 100} if model.dims.n_vocab >= 51866 else {}))
    g = torch.Generator(device=dev).manual_seed(4321)
    pcm = torch.randn((n_per, 480000), generator=g, device=dev) * 0.1
    transcripts = [e2e_transcript(tokenizer, 100 + k) for k in range(n_per)]
    jobs = [WindowJob(pcm[k % n_per], transcripts[k % n_per], 480000, tag=k) for k in range(n_per * steps)]
    out = {}

    def timed(aligner):
        list(align_windows(aligner, jobs[:n_per], n_per))                    # warm-up (allocations, GEMM plans)
        torch.cuda.synchronize()
        aligner.timeline = []
        t0 = time.perf_counter()
        res = list(align_windows(aligner, jobs, n_per))
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        tl = aligner.timeline
        aligner.timeline = None
        stage = {k: float(np.mean([t[k] for t in tl])) for k in tl[0]} if tl else {}
        span_ms = stage.pop("span", 0.0)                     # first launch -> last kernel of a launch set, idle gaps included
        gpu_ms = sum(stage.values())                         # the kernels alone (an event pair around each stage)
        align_ms = sum(v for k, v in stage.items() if k != "model")
        n_words = sum(len(r.words) for r in res)
        assert all(len(r.words) > 0 for r in res) and n_words > 0
        return res, {"audio_s_per_s": round(30.0 * len(jobs) / el, 1), "ms_per_launch_set": round(el / steps * 1e3, 3),
                     "gpu_kernel_ms_per_launch_set": round(gpu_ms, 3),
                     "gpu_stage_ms": {k: round(v, 3) for k, v in stage.items()},
                     "alignment_share_of_gpu_time": round(align_ms / gpu_ms, 4) if gpu_ms else None,
                     "gpu_span_ms_per_launch_set": round(span_ms, 3),
                     "gpu_busy_fraction_of_wall": round(min(1.0, gpu_ms * steps / (el * 1e3)), 4),
                     "words_per_launch_set": n_words // steps}

    opts = dict(language="en", alignment_heads=torch.tensor(heads), refine_whisper_precision_nframes=25)
    if leg == "fp32":
        out = {"workload": f"whisper-{name} (random init, fp32), {n_per} x 30 s synthetic chunks per launch set, "
                           f"{len(transcripts[0])} window tokens in {len(E2E_SEGMENTS)} timestamped segments, teacher forced "
                           f"(naive strategy, trust_whisper_timestamps=False shape)", "chunks_per_launch": n_per,
               "launch_sets": steps, "alignment_heads": len(heads),
               "what_this_leg_is": "the batched SECOND PASS of the naive strategy (naive_approach=True, trust_whisper_timestamps=False: "
                                   "the transcript is given, the decoder is teacher forced) -- not what transcribe(model, audio) does "
                                   "by default; that is the `default_strategy` object below"}
        res32, fp32 = timed(BatchedAligner(model, tokenizer, **opts))
        out.update(fp32)
        out["dtype"] = "f32 model (the CPU reference's arithmetic), f32 alignment, f64 DTW"
        emit(out)
        if not args.no_cpu_baseline:
            # the same chunks through the reference-shaped CPU path, bounded sample
            model_cpu = W.build_model(name, seed=0, device="cpu")
            pcm_cpu = pcm[:8].cpu()
            done, worst_t, worst_c, worst_l, t0 = 0, 0.0, 0.0, 0.0, time.perf_counter()
            while done < 8:
                ws = e2e_cpu_reference_window(W, model_cpu, tokenizer, heads, pcm_cpu[done], transcripts[done])
                got = res32[done]
                assert [x["text"] for x in got.words] == [x["text"] for x in ws], "GPU and CPU words differ"
                for a, lp, b in zip(got.words, got.word_logprobs, ws):
                    worst_t = max(worst_t, abs(a["start"] - b["start"]), abs(a["end"] - b["end"]))
                    conf = lp.mean().exp().item() if len(lp) else 0.0
                    worst_c = max(worst_c, abs(conf - b["confidence_raw"]))
                    # (a random-init model gives p ~ 1/V: the confidences are ~1e-8 and their difference says nothing;
                    #  the mean log-probabilities they are the exp() of are compared as well)
                    assert (len(lp) == 0) == (b["mean_logprob_raw"] is None)
                    if len(lp):
                        worst_l = max(worst_l, abs(lp.mean().item() - b["mean_logprob_raw"]))
                done += 1
                if time.perf_counter() - t0 > args.e2e_cpu_budget:
                    break
            el = time.perf_counter() - t0
            out["cpu_baseline_e2e"] = {"value": round(30.0 * done / el, 2), "unit": "audio-seconds/s",
                                       "cores": int(torch.get_num_threads()), "kind": "port",
                                       "sample": f"{done} of the same chunks, one at a time as the reference does: torch.stft log-mel, "
                                                 f"the same whisper-base on the CPU with unfused attention and per-layer QK capture, "
                                                 f"log_softmax of the (T, V) block, oracle perform_word_alignment, {el:.1f} s wall"}
            out["parity_vs_cpu_reference_path"] = {"chunks": done, "max_abs_dt_word_s": round(worst_t, 4),
                                                   "max_abs_dconfidence_before_rounding": float(f"{worst_c:.3g}"),
                                                   "max_abs_dmean_logprob_per_word": float(f"{worst_l:.3g}"),
                                                   "bars": {"dt_word_s": 0.02, "dconfidence": 1e-4, "dmean_logprob": 2e-4}}
            assert worst_t <= 0.02 + 1e-9 and worst_c <= 1e-4 and worst_l <= 2e-4, out["parity_vs_cpu_reference_path"]
            out["speedup_vs_cpu_e2e"] = round(out["audio_s_per_s"] / out["cpu_baseline_e2e"]["value"], 1)
            emit(out)
        return out
    # the reference's GPU default is fp16=True (transcribe.py:240-241): the same pipeline with half-precision
    # activations -- whisper keeps LayerNorm in fp32 and casts the other weights per call; here they are cast once.
    for m in model.modules():
        if isinstance(m, (torch.nn.Linear, torch.nn.Conv1d, torch.nn.Embedding)):
            m.half()
    res16, fp16 = timed(BatchedAligner(model, tokenizer, mel_dtype=torch.float16, **opts))
    out["fp16_model"] = fp16
    emit(out)
    return out


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="kfull", choices=sorted(WORKLOADS) + ["e2e_base32"],
                    help="kfull (default; its line also carries the transcribe()-level leg), the secondary kernel-level workloads, "
                         "or e2e_base32 = kfull with the e2e leg forced on")
    ap.add_argument("--min-seconds", type=float, default=1.0,
                    help="the --steps region is repeated until this much time has been measured (>= 5 regions)")
    ap.add_argument("--repeats", type=int, default=0, help="fixed number of timed regions (0 = from --min-seconds)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--e2e", default="auto", choices=["auto", "on", "off"],
                    help="transcribe()-level legs (whisper-base, batched.py); auto = with the default workload at N=1")
    ap.add_argument("--other-configs", default="on", choices=["on", "off"],
                    help="N=1, default workload: also run the kreal / kfull256 / largev3_fp16 kernel-level legs (other_configs)")
    ap.add_argument("--e2e-steps", type=int, default=6, help="launch sets of 32 chunks in the e2e timed region")
    ap.add_argument("--e2e-model", default="base", help="shapes of the e2e leg's model (whisper_double names: base = the "
                                                         "BASELINE config; small, medium, large-v3 ... for other shapes)")
    ap.add_argument("--e2e-windows", type=int, default=32, help="30 s chunks per launch set of the e2e leg")
    ap.add_argument("--e2e-cpu-budget", type=float, default=15.0)
    ap.add_argument("--graph", action="store_true",
                    help="replay the step as ONE captured HIP graph (fixed shapes); stage times then come from an eager "
                         "pass BEFORE the timed region (events cannot be read back from inside a graph)")
    ap.add_argument("--gather-every", type=int, default=8,
                    help="N>1: result records of this many steps travel to rank 0 in one RCCL gather")
    ap.add_argument("--pipeline", type=int, default=2,
                    help="independent batches in flight: step k runs on HIP stream k %% N with its own output buffers (the "
                         "inputs are shared), so the 32-CU, latency-bound DTW of one step overlaps the other steps' kernels "
                         "with no cross-stream dependency at all.  The line also carries the single-batch-in-flight time; "
                         "per-stage times and the roofline always come from the single-stream pass")
    ap.add_argument("--schedule", default="auto", choices=sorted(SCHEDULES) + ["auto"],
                    help="how the stages of one batch share the chip when --pipeline > 1 (see SCHEDULES): serial = one stream per "
                         "batch; hilo = per batch a high-priority stream (stft_mel, dtw_kernel) and a low-priority one (cost, log-prob); "
                         "auto = hilo where the DTW leaves most of the chip idle (<= 128 units per step, split cost / DTW entries), "
                         "serial for the 256-unit and the fused small-unit workloads (measured: profiles/r5f_bench_driver_command.json "
                         "vs r4u_bench_final.json)")
    ap.add_argument("--align", default="auto", choices=["auto", "split", "fused"],
                    help="split: wt_cost_batch then wt_dtw_batch (two timed stages, batched kernels only); fused: ONE "
                         "wt_align_batch_v3 (small units through the fused kernel; timed as the cost stage); auto = fused for "
                         "the workloads that have small units (kreal)")
    ap.add_argument("--dtw-cus", type=int, default=32, help="--overlap cumask: CUs reserved for the DTW stream")
    ap.add_argument("--overlap", default="none", choices=["none", "lanes", "dtw", "dtw_logmel", "cumask"],
                    help="none: one stream; lanes: log-mel | cost+DTW | log-prob on three HIP streams; "
                         "dtw: only the (32-CU, latency-bound) DTW runs beside the (HBM-bound) log-prob gather")
    # --- process plumbing (see orchestrate()): the measuring legs run in child processes of this script
    ap.add_argument("--role", default="orchestrate", choices=["orchestrate", "kernel", "cpu", "e2e"], help=argparse.SUPPRESS)
    ap.add_argument("--leg", default="fp32", choices=["fp32", "fp16", "efficient", "recordings"], help=argparse.SUPPRESS)
    ap.add_argument("--e2e-streams", type=int, default=32,
                    help="recordings per decoder op of the default-strategy leg (transcribe_batch; 32 = BASELINE configs[1])")
    ap.add_argument("--e2e-worker-processes", type=int, default=0,
                    help="default-strategy leg: worker processes of the ragged long-form sub-leg (0 / 1 = skip it, the default: "
                         "measured once, profiles/r5h_*: 4 processes x 8 streams = 1.33x one process x 32 streams -- the "
                         "processes' small kernels serialise on the one device)")
    ap.add_argument("--out", default=None, help=argparse.SUPPRESS)
    ap.add_argument("--secondary", action="store_true", help=argparse.SUPPRESS)     # a kernel leg of another BASELINE config
    ap.add_argument("--dry-run", action="store_true",
                    help="launcher / process-group plumbing only (gloo on the CPU, no kernels, meaningless numbers): what "
                         "tests/test_bench_launcher.py runs where there is no GPU")
    ap.add_argument("--inject-fault", default="", help=argparse.SUPPRESS)   # tests: "kernel", "e2e_fp16" ... abort that child
    args = ap.parse_args(argv)
    if args.workload == "e2e_base32":
        args.workload, args.e2e = "kfull", "on"
    if args.overlap != "none":
        args.pipeline = 1
    return args


def _json_scalar(o):
    """numpy scalars that slipped into a result dictionary"""
    if hasattr(o, "item"):
        return o.item()
    raise TypeError(f"Object of type {o.__class__.__name__} is not JSON serializable")


def make_emitter(path):
    """Children publish their (partial) results by atomically rewriting one JSON file: whatever was measured before
    a GPU fault is still there for the parent."""
    def emit(obj):
        if not path:
            return
        tmp = path + ".tmp"
        with open(tmp, "w") as f:
            json.dump(obj, f, default=_json_scalar)
        os.replace(tmp, path)
    return emit


def role_kernel(args):
    """The kernel-level measurement (one process per GPU).  Publishes the single-batch-in-flight line as soon as it
    exists, then the line with `--pipeline` batches in flight."""
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

    cfg = WORKLOADS[args.workload]
    n = cfg["n_chunks"]
    if dry:
        w = dict(cfg=cfg, jumps=torch.zeros(n * (cfg["T"] + 1), dtype=torch.int32), logprob=torch.zeros(n * cfg["T"]))
    else:
        w = make_workload(dev, cfg, seed=1234 + rank)
        w["align"] = args.align if args.align != "auto" else ("fused" if cfg.get("units_per_chunk") else "split")
    cfg = w["cfg"]

    gatherers = None
    if world > 1 or force_dist:
        from whisper_timestamped.sharding import ResultGatherer
        gatherers = [ResultGatherer(dist, w["jumps"].numel(), w["logprob"].numel(), dev, every=args.gather_every)
                     for _ in range(args.pipeline)]
    gather_buf = gatherers[0] if gatherers else None

    streams = None
    if args.overlap == "lanes":
        streams = [torch.cuda.Stream(device=dev) for _ in LANES]
    elif args.overlap in ("dtw", "dtw_logmel"):
        streams = args.overlap, torch.cuda.Stream(device=dev)
    elif args.overlap == "cumask":
        streams = ("cumask",) + cu_masked_streams(dev, args.dtw_cus)

    # --pipeline N: N output-buffer sets over the same inputs, one stream each
    pipe = [w]
    pipe_streams = [None]
    if args.pipeline > 1 and not dry:
        n_cost = w["cost"].numel()
        for _ in range(args.pipeline - 1):
            c = dict(w)
            c.pop("_calls", None)
            c.update(cost=torch.empty(n_cost, dtype=torch.float32, device=dev), mel=torch.empty_like(w["mel"]),
                     gmax=torch.empty_like(w["gmax"]), pad=torch.empty_like(w["pad"]),
                     **result_buffers(w["jumps"].numel(), w["logprob"].numel(), dev))
            pipe.append(c)
        pipe_streams = [torch.cuda.Stream(device=dev) for _ in range(args.pipeline)]
    schedule = args.schedule
    if schedule == "auto":
        schedule = "hilo" if (not dry and w.get("align") == "split" and len(w["descs"]) <= 128) else "serial"
    plan = SCHEDULES[schedule] if (args.pipeline > 1 and not dry and not args.graph) else None
    if plan is not None and w.get("align") == "fused":
        plan = dict(plan, order=[st_ for st_ in plan["order"] if st_ != "dtw"])      # (one entry point: cost + DTW as the cost stage)
    shared_streams = {}
    schedule_note = None
    plan_stream_sets = None
    if plan is not None:
        try:
            plan_stream_sets = [plan_streams(dev, plan, shared_streams) for _ in range(args.pipeline)]
        except Exception as e:                             # noqa: BLE001 -- a runtime without stream priorities: the round-4 schedule
            schedule_note = f"{schedule} not available ({e!r}): serial"
            print(f"[bench] {schedule_note}", file=sys.stderr, flush=True)
            plan, schedule = None, "serial"

    rank_seconds = []            # N > 1: per timed region, every rank's own seconds (before the closing barrier)
    use_gather = [True]          # (switched off for the "what does the gather cost" regions at the end)

    def full_step(ev=None, k=0, pipelined=False):
        if dry:
            time.sleep(2e-4)
            if gatherers is not None and use_gather[0]:
                gatherers[k % args.pipeline if pipelined else 0].gather(w["jumps"], w["logprob"])
            return
        if pipelined:
            j = k % args.pipeline
            if plan is not None:
                last = run_step_plan(pipe[j], plan, plan_stream_sets[j])
                if gatherers is not None and use_gather[0]:
                    with torch.cuda.stream(last):
                        gatherers[j].gather(pipe[j]["jumps"], pipe[j]["logprob"])
                return
            with torch.cuda.stream(pipe_streams[j]):
                run_step(pipe[j], ev, None)
                if gatherers is not None and use_gather[0]:
                    gatherers[j].gather(pipe[j]["jumps"], pipe[j]["logprob"])
            return
        run_step(w, ev, streams)
        if gather_buf is not None and use_gather[0]:
            gather_buf.gather(w["jumps"], w["logprob"])

    for k in range(args.warmup):
        full_step(None, k)
    sync()
    if args.pipeline > 1:
        for k in range(max(args.warmup, 2 * args.pipeline)):       # every stream's scratch arenas exist before the timing
            full_step(None, k, pipelined=True)
        if gatherers is not None:
            for g_ in gatherers:
                g_.drain()
        sync()

    graph = None
    if args.graph:
        # ONE captured HIP graph holding `pipeline` steps: batch 0 on the capture stream, every other batch on a branch
        # forked at the head of the graph and joined at its end (independent branches: the runtime may run them side by
        # side, as the eager two-stream pipeline does, without the per-launch host cost)
        assert args.overlap == "none" and gather_buf is None and not dry, "--graph: single rank"
        cap = torch.cuda.Stream(device=dev)
        sides = [torch.cuda.Stream(device=dev) for _ in range(args.pipeline - 1)]
        cap.wait_stream(torch.cuda.current_stream())
        for j, s_ in enumerate([cap] + sides):       # the library's scratch arenas are per stream: create them
            with torch.cuda.stream(s_):              # (hipMalloc) before the capture, not inside it
                run_step(pipe[j], None, None)
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=cap):
            for s_ in sides:
                s_.wait_stream(cap)                  # fork
            run_step(pipe[0], None, None)
            for j, s_ in enumerate(sides, 1):
                with torch.cuda.stream(s_):
                    run_step(pipe[j], None, None)
            for s_ in sides:
                cap.wait_stream(s_)                  # join
        graph.replay()
        torch.cuda.synchronize()

    def make_events():
        if dry:
            return None
        if args.overlap != "none":
            return {st: (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for st in STAGES}
        order = [stage for lane in LANES for stage in lane]
        marks = [torch.cuda.Event(enable_timing=True) for _ in range(len(order) + 1)]
        return {st: (marks[i], marks[i + 1]) for i, st in enumerate(order)}

    evs = [make_events() for _ in range(args.steps)]
    stage_samples = {s: [] for s in STAGES}

    def timed_region(pipelined=False):
        """EXACTLY args.steps steps between barrier + synchronize on both sides; max over ranks; seconds."""
        if dist is not None:
            dist.barrier()
        sync()
        t0 = time.perf_counter()
        if graph is not None and (pipelined or args.pipeline == 1):
            reps, rem = divmod(args.steps, args.pipeline)     # one replay = `pipeline` steps
            for k in range(reps):
                graph.replay()
            for k in range(rem):
                full_step(None, k, args.pipeline > 1)
        else:
            for k in range(args.steps):
                full_step(None if pipelined else evs[k], k, pipelined)
        if gatherers is not None:
            for g_ in (gatherers if pipelined else gatherers[:1]):
                g_.drain()
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
        elif not pipelined and (graph is None or args.pipeline > 1):
            for s in STAGES:
                stage_samples[s].extend(evs[k][s][0].elapsed_time(evs[k][s][1]) for k in range(args.steps))
        return el

    if graph is not None and args.pipeline == 1:     # eager pass for the per-stage breakdown (NOT part of the timing)
        for k in range(args.steps):
            full_step(evs[k])
        torch.cuda.synchronize()
        for s in STAGES:
            stage_samples[s].extend(evs[k][s][0].elapsed_time(evs[k][s][1]) for k in range(args.steps))

    def measure(pipelined):
        regions = [timed_region(pipelined)]
        n_regions = args.repeats or int(min(2000, max(5, np.ceil(args.min_seconds / max(regions[0], 1e-6)))))
        while len(regions) < n_regions:              # (every rank derives the same count from the max-reduced first region)
            regions.append(timed_region(pipelined))
        return regions

    def check_results(sets):
        """sanity inside the bench: the ridge is recovered and log-probs are finite (every buffer set given)"""
        if dry:
            return
        torch.cuda.synchronize()
        for c in sets[1:]:
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
        ab = algorithmic_bytes(cfg, w.get("align") == "fused")
        dom = max(stage_ms, key=stage_ms.get)
        achieved = ab[dom] / (stage_ms[dom] * 1e-3) / 1e9
        stages = {s: {"ms": round(stage_ms[s], 4), "alg_MB": round(ab[s] / 1e6, 2),
                      "GBps": round(ab[s] / (stage_ms[s] * 1e-3) / 1e9, 1),
                      "frac_hbm": round(ab[s] / (stage_ms[s] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)} for s in STAGES}
        traffic, traffic_src = committed_traffic(dom, args.workload)
        return {
            "metric": "audio-seconds aligned/sec (whole node), whisper-base 30s chunks",
            "value": round(world * n * 30.0 * args.steps / elapsed, 1),
            "unit": "audio-seconds/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
            "data": "synthetic" if not dry else "DRY RUN: no kernels ran, the numbers mean nothing",
            "config": {"workload": cfg["desc"], "units_per_step_per_gpu": n, "units_per_step": len(w["descs"]) if not dry else n,
                       "padded_units": f"1 chunk in {PADDED_EVERY} ends in silence (pad_from = U[F/2, F), its PCM zero from there): the "
                                       "padding detector and the pad mask run inside the timed region",
                       "stages": STAGES,
                       "arithmetic": "f32 cost / log-softmax / log-mel (as the reference's torch CPU ops), f64 DTW (as dtw-python)",
                       "dtw_oracle": "published dtw-python algorithm (symmetric1, strict-< tie order), unpinned against the "
                                     "package itself: absent from the image (tests/test_oracle.py pins it on exhaustive "
                                     "path enumeration and on transformers' DTW for tie-free inputs)",
                       "streams": {"none": 1, "lanes": 3, "dtw": 2, "dtw_logmel": 2, "cumask": 3}[args.overlap],
                       "hip_graph": bool(args.graph), "batches_in_flight": batches_in_flight,
                       "schedule": (schedule if plan is not None and batches_in_flight > 1 else "serial"),
                       "schedule_streams": ({k_: list(v_) for k_, v_ in plan["assign"].items()} if plan is not None and batches_in_flight > 1 else None),
                       "schedule_note": schedule_note,
                       "alignment_entry": "wt_align_batch_v3 (batched row pass + fused small-unit tail kernel; timed as the cost stage)"
                                          if w.get("align") == "fused" else "wt_cost_batch + wt_dtw_batch",
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
            "stages": stages,
            **extras,
        }

    single_regions = measure(False)                  # one batch in flight: also the per-stage times and the roofline
    check_results(pipe[:1])
    if not dry:
        # a few units of what the timed region has just computed, against the oracle (every rank checks its own batch)
        extras["parity_in_leg"] = parity_in_leg(w)
        assert extras["parity_in_leg"]["ok"], extras["parity_in_leg"]
    if rank == 0:
        emit(line(single_regions, single_regions, 1))     # published before the multi-stream pass starts
    mark = len(rank_seconds)
    if args.pipeline > 1 and not dry:
        # the parity check above kept the GPU idle for seconds (the oracle runs on the host): the --warmup steps again, on the
        # pipelined path, before its regions are timed
        for k in range(max(args.warmup, 2 * args.pipeline)):
            full_step(None, k, pipelined=True)
        if gatherers is not None:
            for g_ in gatherers:
                g_.drain()
        sync()
    regions = measure(True) if args.pipeline > 1 else single_regions
    check_results(pipe)
    if dist is not None and world > 1:
        # N > 1: what every rank needed for the same region (a bad scaling curve can be read: one slow GPU, or all of
        # them waiting), and what the result gather to rank 0 costs (the same regions once more without it)
        per = np.median(np.asarray(rank_seconds[mark:] if args.pipeline > 1 else rank_seconds), axis=0) / args.steps * 1e3
        extras["per_rank"] = {"ms_per_step": [round(float(x), 4) for x in per], "min": round(float(per.min()), 4),
                              "max": round(float(per.max()), 4), "skew_max_over_min": round(float(per.max() / per.min()), 4),
                              "note": "each rank's own time from the opening barrier to its last kernel / gather done, "
                                      "median over the timed regions; the headline is the max over ranks incl. the closing barrier"}
        if gatherers is not None:
            use_gather[0] = False
            bare = [timed_region(args.pipeline > 1) for _ in range(5)]
            use_gather[0] = True
            extras["result_gather"] = {"ms_per_step_without_gather": round(float(np.median(bare)) / args.steps * 1e3, 4),
                                       "share_of_step": round(max(0.0, 1.0 - float(np.median(bare)) / float(np.median(regions))), 4)}
    if rank == 0:
        emit(line(regions, single_regions, args.pipeline))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def role_cpu(args):
    """cpu_baseline: the oracle on the host cores over a bounded sample of the same workload (the sample is drawn on
    the GPU with the kernel leg's generator, then moved to the host)."""
    emit = make_emitter(args.out)
    dev = torch.device("cuda", 0)
    cfg = WORKLOADS[args.workload]
    w = make_workload(dev, cfg, seed=1234)
    out = {"cpu_baseline": cpu_baseline(w["cfg"], w)}
    emit(out)
    out["cpu_baseline_1thread"] = cpu_baseline(w["cfg"], w, budget_s=8.0, threads=1)
    emit(out)


def ragged_window(rs, frames, ts0, eot, lo=40, hi=160):
    """tests/many_helper.ragged_window (the scripts are also built inside worker processes)."""
    import many_helper as H
    return H.ragged_window(rs, frames, ts0, eot, lo, hi)


def words_of(r):
    return [(w["text"], w["start"], w["end"], w["confidence"]) for s_ in r["segments"] for w in s_["words"]]


def word_gaps(a, b, what):
    """a, b: words_of() of two runs with RAW confidences (words.RAW_CONFIDENCE).  -> [max |dt|, max |dconfidence|,
    max |d mean log-prob|, words, words whose start or end differs by more than 0.02 s]: confidence = exp(mean log-prob
    of the word's tokens), so log(confidence) IS the mean."""
    import math
    assert [x[0] for x in a] == [x[0] for x in b], f"{what}: words differ"
    dts = [max(abs(x[1] - y[1]), abs(x[2] - y[2])) for x, y in zip(a, b)]
    dc = max([0.0] + [abs(x[3] - y[3]) for x, y in zip(a, b)])
    dl = 0.0
    for x, y in zip(a, b):
        assert (x[3] == 0) == (y[3] == 0), (what, x, y)
        if x[3] and y[3]:
            dl = max(dl, abs(math.log(x[3]) - math.log(y[3])))
    return [float(max([0.0] + dts)), float(dc), float(dl), len(dts), int(sum(d > 0.02 + 1e-9 for d in dts))]


def merge_gaps(worst, new):
    return [max(worst[0], new[0]), max(worst[1], new[1]), max(worst[2], new[2]), worst[3] + new[3], worst[4] + new[4]]


NO_GAPS = [0.0, 0.0, 0.0, 0, 0]
# B streams against ONE stream of the same recording: the alignment kernels are deterministic and batch-independent
# (tests/test_gpu_parity.py::test_cost_and_jumps_do_not_depend_on_the_batch), but the backend's GEMMs are not bit-identical
# between a batch of 32 and a batch of 1 (other tile shapes, other accumulation orders: ~1e-6 relative in q and K).  A
# random-init model's cross-attention is nearly flat, so where the script repeats a token the DTW has near-ties and that
# noise can move a boundary locally (profiles/r5c_diag_ragged_parity.txt: 4 of 1070 words, one recording, confidences
# identical to 2e-6; the streams driver run ONE stream at a time equals transcribe() word for word).  Asserted: texts,
# confidences and mean log-probabilities for every word, times within 0.02 s for at least 99 % of the words; the count and
# the worst gap are reported.
MAX_SHARE_OF_WORDS_MOVED_BY_BATCH_ROUNDING = 0.01


PARITY_FAILURES = []      # legs whose parity check did not hold: reported in the line (`parity_failures`), never hidden


def parity_flag(ok, what, detail):
    """A parity check of a transcribe()-level sub-leg: recorded, logged, and the leg goes on (an assert here would cost
    every sub-leg behind it); the line carries every failure at its top level."""
    if not ok:
        PARITY_FAILURES.append({"leg": what, "detail": detail})
        print(f"[bench] PARITY CHECK FAILED in {what}: {detail}", file=sys.stderr, flush=True)
    return bool(ok)


def gaps_report(worst, extra=None):
    rep = dict(extra or {})
    rep.update({"words_compared": worst[3], "words_beyond_0.02_s": worst[4], "max_abs_dt_word_s": round(float(worst[0]), 4),
                "max_abs_dconfidence_before_rounding": float(f"{worst[1]:.3g}"), "max_abs_dmean_logprob_per_word": float(f"{worst[2]:.3g}")})
    return rep


def gaps_ok_between_batch_sizes(worst):
    return worst[1] <= 1e-4 and worst[2] <= 2e-4 and worst[4] <= max(1, MAX_SHARE_OF_WORDS_MOVED_BY_BATCH_ROUNDING * worst[3])


def run_efficient_leg(args, emit):
    """The DEFAULT strategy of transcribe() (the reference's efficient strategy: word alignment on the fly while the
    backend decodes, T.py:359-1001), whisper double as the model:
      1_stream          what a caller of the reference's API gets per process: transcribe(model, clip), one decoder stream,
                        one token at a time through the backend's own Python loop;
      B_streams         transcribe_batch(model, clips): B independent recordings stepping through the decoder together
                        (whisper_timestamped/streams.py), B = --e2e-streams (32 = BASELINE configs[1]'s batch), and 4 B --
                        UNIFORM work: 30 s clips, one scripted ~110-token transcript in 5 segments for every stream (every
                        stream finishes in the same decoder call: the lock-step best case);
      ragged_B_streams  the same on RAGGED work: clip lengths U[5, 30] s, a different scripted transcript per stream
                        (2-9 segments, 40-160 tokens), with the driver's streams-per-loop histogram;
      long_form_1h_islands  BASELINE configs[3] at N = 1, uniform and ragged (per-window transcripts drawn per island, so
                        the prompts of windows 2, 3 differ in length from stream to stream under condition_on_previous_text);
      cpu_baseline      the reference-shaped CPU path for the same clips: the same model on the host cores, unfused
                        attention with per-token QK capture, a second projection + logit filters per token, one
                        synchronous alignment per segment through oracle/ (the reference's shape, T.py:783-793,849-881,
                        544-557), one stream -- a bounded sample;
      parity            every B-stream recording against the one-stream output (word times, raw confidences, mean
                        log-probabilities) and the sampled clips against the CPU path's."""
    import many_helper as H          # tests/: the whisper double as the model, the scripted transcript
    import whisper_double as W
    from whisper_double.decoding import Script, set_row_scripts, set_script
    from golden import make_golden_transcribe as G
    W.install()
    import whisper_timestamped as wt
    from whisper_timestamped import streams, words
    words.RAW_CONFIDENCE = True      # confidences before the reference's round(, 3): parity is asserted on the raw values
    dev = getattr(args, "e2e_device", "cuda:0")     # (a CPU dry run of this leg's host logic: tools/dry_run_efficient_leg.py)
    model = H.load_base(dev)
    B = args.e2e_streams
    TS0, EOT = 50364, 50257
    g = torch.Generator().manual_seed(7)
    clips = [(0.05 * torch.randn(30 * 16000, generator=g)).float() for _ in range(4)]
    segs = [(s, [None] * n, e) for s, n, e in H.SEGMENTS]
    window = G.window_script(TS0, EOT, segs, "eot")
    out = {"workload": "whisper-base (random init, fp32), synthetic clips, scripted transcripts, transcribe() with its defaults "
                       "(efficient strategy, greedy, condition_on_previous_text=True); uniform legs: 30 s clips, one ~110-token "
                       "transcript in 5 timestamped segments for every stream; ragged legs: U[5, 30] s clips, 2-9 segments and "
                       "40-160 tokens drawn per stream"}
    bars = {"dt_word_s": 0.02, "dconfidence": 1e-4, "dmean_logprob": 2e-4}

    # ---- one stream (the reference's shape of the call)
    def one(clip, windows=None, **kw):
        set_script(Script(windows if windows is not None else [window]))
        try:
            return wt.transcribe(model, clip, language="en", fp16=False, **kw)
        finally:
            set_script(None)
    one(clips[0])                                           # warm-up: allocations, GEMM plans, the library's arenas
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    singles = [one(c) for c in clips]
    torch.cuda.synchronize()
    el1 = time.perf_counter() - t0
    n_words = sum(len(words_of(r)) for r in singles)
    assert n_words > 0
    out["1_stream"] = {"audio_s_per_s": round(30.0 * len(clips) / el1, 1), "clips": len(clips), "seconds": round(el1, 3),
                       "ms_per_clip": round(1e3 * el1 / len(clips), 1), "words": n_words}
    emit(out)

    # ---- B streams per decoder op
    def batch_of(audios, window_lists, max_streams, **kw):
        scripts = [Script(ws) for ws in window_lists]

        def on_group(idx):
            for i in idx:
                scripts[i].begin_window()
            set_row_scripts([scripts[i] for i in idx])
        streams.ON_GROUP_DECODE = on_group
        try:
            return wt.transcribe_batch(model, audios, max_streams=max_streams, language="en", fp16=False, **kw)
        finally:
            streams.ON_GROUP_DECODE = None
            set_row_scripts(None)

    def many(n):
        return batch_of([clips[k % len(clips)] for k in range(n)], [[window]] * n, n)

    def histogram(sizes):
        h = {}
        for x in sizes:
            h[int(x)] = h.get(int(x), 0) + 1
        return {str(k): h[k] for k in sorted(h)}

    def driver_stats():
        d = dict(streams.LAST_RUN)
        sizes = d.pop("streams_per_loop", [])
        d["streams_per_loop_histogram"] = histogram(sizes)
        d["mean_streams_per_loop"] = round(float(np.mean(sizes)), 2) if sizes else None
        return d

    for n_streams in (B, 4 * B):
        many(n_streams)                                     # warm-up at the timed shape
        torch.cuda.synchronize()
        reps = 3
        t0 = time.perf_counter()
        for _ in range(reps):
            batch = many(n_streams)
        torch.cuda.synchronize()
        elB = (time.perf_counter() - t0) / reps
        worst = NO_GAPS
        for k, r in enumerate(batch):
            worst = merge_gaps(worst, word_gaps(words_of(r), words_of(singles[k % len(clips)]), "B-stream vs one-stream"))
        key = f"{n_streams}_streams"
        out[key] = {"audio_s_per_s": round(30.0 * n_streams / elB, 1), "clips": n_streams, "seconds": round(elB, 3),
                    "ms_per_clip": round(1e3 * elB / n_streams, 2), "words": sum(len(words_of(r)) for r in batch),
                    "speedup_vs_1_stream": round((30.0 * n_streams / elB) / (30.0 * len(clips) / el1), 2),
                    "driver": driver_stats(), "parity_vs_1_stream": gaps_report(worst)}
        out[key]["parity_vs_1_stream"]["ok"] = parity_flag(gaps_ok_between_batch_sizes(worst), key, out[key]["parity_vs_1_stream"])
        emit(out)

    # ---- the same on RAGGED work: clip lengths U[5, 30] s, a different transcript per stream
    def ragged_jobs(n, seed):
        rs = np.random.RandomState(seed)
        audios, wins, secs = [], [], []
        for k in range(n):
            sec = float(rs.uniform(5.0, 30.0))
            audios.append(clips[k % len(clips)][:int(sec * 16000)].clone())
            wins.append([ragged_window(rs, int(sec * 50), TS0, EOT)])
            secs.append(sec)
        return audios, wins, secs
    for n_streams in (B, 4 * B):
        print(f"[bench] default strategy: ragged, {n_streams} streams", file=sys.stderr, flush=True)
        audios, wins, secs = ragged_jobs(n_streams, 100 + n_streams)
        batch_of(audios, wins, n_streams)                   # warm-up at the timed shape
        torch.cuda.synchronize()
        reps = 3
        t0 = time.perf_counter()
        for _ in range(reps):
            batch = batch_of(audios, wins, n_streams)
        torch.cuda.synchronize()
        elR = (time.perf_counter() - t0) / reps
        stats = driver_stats()
        worst, checked = NO_GAPS, 0
        for k in range(0, n_streams, max(1, n_streams // 16)):          # 16 of the recordings, one stream at a time
            worst = merge_gaps(worst, word_gaps(words_of(batch[k]), words_of(one(audios[k], wins[k])), "ragged B-stream vs one-stream"))
            checked += 1
        key = f"ragged_{n_streams}_streams"
        tok = [len(w_[0]) for w_ in wins]
        out[key] = {"audio_s_per_s": round(sum(secs) / elR, 1), "clips": n_streams, "audio_seconds": round(sum(secs), 1),
                    "clip_seconds": "U[5, 30]", "tokens_per_transcript": {"min": min(tok), "mean": round(float(np.mean(tok)), 1), "max": max(tok)},
                    "seconds": round(elR, 3), "words": sum(len(words_of(r)) for r in batch), "driver": stats,
                    "parity_vs_1_stream": gaps_report(worst, {"recordings_compared": checked})}
        out[key]["parity_vs_1_stream"]["ok"] = parity_flag(gaps_ok_between_batch_sizes(worst), key, out[key]["parity_vs_1_stream"])
        emit(out)

    # ---- BASELINE configs[3] at N = 1: ONE long recording (1 h), its speech islands given (the reference's vad=[...] form;
    #      silero itself needs network), every island an independent unit -> the rank's islands as decoder streams.  On N
    #      ranks the same call deals the islands to the ranks first (sharding.transcribe_islands, no data-path collective).
    from whisper_timestamped.sharding import transcribe_islands
    pattern = (90, 30, 30, 60, 30, 60)                      # island lengths in seconds: one to three 30 s windows each
    n_islands = getattr(args, "e2e_islands", 72)                         # 72: 12 x 300 s = one hour
    durations = [pattern[k % len(pattern)] for k in range(n_islands)]
    total_s = sum(durations)
    assert total_s == 3600 or n_islands != 72
    hour = torch.cat([clips[k % len(clips)] for k in range(total_s // 30)])
    islands, t = [], 0.0
    for d_ in durations:
        islands.append((t, t + d_))
        t += d_
    ragged_island_windows = H.ragged_island_windows(durations, seed=77, ts0=TS0, eot=EOT)
    uniform_island_windows = [[window] * (d_ // 30) for d_ in durations]
    n_windows = sum(d_ // 30 for d_ in durations)
    out["long_form_1h_islands"] = {
        "islands": len(islands), "island_seconds": "30 / 60 / 90 (one to three windows each)", "windows": n_windows,
        "streams_per_decoder_op": B,
        "note": "BASELINE configs[3] at N = 1: explicit speech islands of one 1 h recording (sharding.transcribe_islands("
                "streams=B)): an island that is finished hands its place to the next one; on N ranks the islands are dealt to "
                "the ranks first.  Streams share a decoder loop only when their prompts have the same LENGTH (the decoder has no "
                "padding mask: padding would move the positions and change the result): with the reference's default "
                "condition_on_previous_text=True the later windows of a recording form their own loops until the prompt "
                "saturates at 223 tokens -- `uniform` scripts one transcript for every window (equal prompt lengths at equal "
                "window index: the best case), `ragged` draws every window's transcript (2-9 segments, 40-160 tokens) per island"}

    def island_run(window_lists, cond, hold=0):
        def on_batch(indices):
            scripts = [Script(window_lists[i]) for i in indices]

            def on_group(rows):                              # rows: positions in the rank's list of islands
                for r in rows:
                    scripts[r].begin_window()
                set_row_scripts([scripts[r] for r in rows])
            streams.ON_GROUP_DECODE = on_group
        streams.HOLD_FOR_BUCKET = hold
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        try:
            merged = transcribe_islands(model, hour, islands, streams=B, on_batch=on_batch, language="en", fp16=False,
                                        condition_on_previous_text=cond)
        finally:
            streams.ON_GROUP_DECODE = None
            streams.HOLD_FOR_BUCKET = 0
            set_row_scripts(None)
        torch.cuda.synchronize()
        return merged, time.perf_counter() - t0

    def island_parity(merged, window_lists, cond, picks):
        """`picks` islands: transcribe() of the island's crop, one stream, against the island's words in the merged result."""
        worst = NO_GAPS
        for i in picks:
            s_, e_ = islands[i]
            crop = hour[int(round(s_ * 16000)):int(round(e_ * 16000))]
            alone = one(crop, window_lists[i], condition_on_previous_text=cond)
            mine = [(w["text"], w["start"] - s_, w["end"] - s_, w["confidence"]) for seg in merged["segments"]
                    if s_ - 1e-6 <= seg["start"] < e_ - 1e-6 for w in seg["words"]]
            worst = merge_gaps(worst, word_gaps(mine, words_of(alone), f"island {i} vs transcribe(crop)"))
        rep = gaps_report(worst, {"islands_compared_with_transcribe_of_the_crop": list(picks)})
        rep["ok"] = parity_flag(gaps_ok_between_batch_sizes(worst), "long_form_1h_islands", rep)
        return rep

    ragged_words_per_island = []
    legs = [("condition_on_previous_text", uniform_island_windows, True, 0), ("no_condition", uniform_island_windows, False, 0),
            ("ragged", ragged_island_windows, True, 0), ("ragged_bucket_admission", ragged_island_windows, True, 2),
            ("ragged_no_condition", ragged_island_windows, False, 0)]
    for label, window_lists, cond, hold in legs:
        print(f"[bench] default strategy, long form: {label}", file=sys.stderr, flush=True)
        merged, el_h = island_run(window_lists, cond, hold)
        stats = driver_stats()
        n_seg_expected = sum(sum(1 for t_ in w_[:-1] if t_ is not None and t_ >= TS0) // 2 for ws in window_lists for w_ in ws)
        assert len(merged["segments"]) == n_seg_expected, (label, len(merged["segments"]), n_seg_expected)
        starts = [s_["start"] for s_ in merged["segments"]]
        assert starts == sorted(starts) and all(len(s_["words"]) > 0 for s_ in merged["segments"])
        rec = {"audio_s_per_s": round(total_s / el_h, 1), "seconds": round(el_h, 3), "driver": stats, "segments": len(merged["segments"]),
               "words": sum(len(s_["words"]) for s_ in merged["segments"]), "condition_on_previous_text": cond}
        if hold:
            rec["hold_for_bucket"] = hold
        if label == "ragged":                                   # (per island, for the worker-process leg below)
            per = []
            for s_, e_ in islands:
                per.append([(w["text"], round(w["start"] - s_, 2), round(w["end"] - s_, 2)) for seg in merged["segments"]
                            if s_ - 1e-6 <= seg["start"] < e_ - 1e-6 for w in seg["words"]])
            ragged_words_per_island[:] = per
        if label in ("condition_on_previous_text", "ragged"):
            rec["parity_vs_1_stream"] = island_parity(merged, window_lists, cond, [i for i in (0, 1, 3, 6, 12) if i < n_islands])
        out["long_form_1h_islands"][label] = rec
        emit(out)

    # ---- what recovers the ragged loss on ONE GPU: processes.  A decoder loop is bound by its one Python thread whether it
    #      carries 1 stream or 32, so W worker processes (own interpreter, own HIP queues, own copy of the 290 MB model) run W
    #      loops side by side: the islands as independent recordings through sharding.transcribe_many(streams=B / W).  On N
    #      GPUs the N ranks ARE such processes.  (Timed between the workers' common start and the last result; process
    #      start-up and model load are reported beside it.)
    if dev != "cpu" and getattr(args, "e2e_worker_processes", 0) > 1:
        import functools
        from whisper_timestamped.sharding import transcribe_many
        W_ = int(getattr(args, "e2e_worker_processes", 0))
        print(f"[bench] default strategy, long form: ragged, {W_} worker processes", file=sys.stderr, flush=True)
        crops = [hour[int(round(s_ * 16000)):int(round(e_ * 16000))].clone() for s_, e_ in islands]
        t0 = time.perf_counter()
        try:
            res_w, slowest = transcribe_many(H.load_base, crops, workers_per_gpu=W_, devices=[dev], warmup=True, return_timing=True,
                                             streams=max(1, B // W_), language="en", fp16=False,
                                             on_batch=functools.partial(H.script_ragged_islands, durations=tuple(durations), seed=77))
            wall = time.perf_counter() - t0
            ref, n_w, n_moved = ragged_words_per_island, 0, 0
            for i, r in enumerate(res_w):                         # (8 streams per loop there, 32 here: batch-size rounding, see word_gaps)
                mine = [(x[0], x[1], x[2]) for x in words_of(r)]
                assert [x[0] for x in mine] == [x[0] for x in ref[i]], f"island {i}: words differ between one process and {W_}"
                n_w += len(mine)
                n_moved += sum(max(abs(a_[1] - b_[1]), abs(a_[2] - b_[2])) > 0.02 + 1e-9 for a_, b_ in zip(mine, ref[i]))
            out["long_form_1h_islands"]["ragged_worker_processes"] = {
                "worker_processes": W_, "streams_per_process": max(1, B // W_), "audio_s_per_s": round(total_s / slowest, 1),
                "seconds": round(slowest, 3), "seconds_incl_process_start_and_model_load": round(wall, 2),
                "words_compared_with_the_one_process_run": n_w, "words_beyond_0.02_s": int(n_moved),
                "vs_one_process": round((total_s / slowest) / out["long_form_1h_islands"]["ragged"]["audio_s_per_s"], 2)}
        except Exception as e:                                   # noqa: BLE001 -- an optional leg must not cost the others
            out["long_form_1h_islands"]["ragged_worker_processes"] = {"error": repr(e)[:300]}
        emit(out)

    # ---- the reference-shaped CPU path, same clips (bounded sample)
    if not args.no_cpu_baseline:
        import cpu_kernel_standin
        from whisper_timestamped import efficient
        saved = {k: getattr(efficient, k) for k in ("REUSE_DECODER_LOGITS", "DEFER_ALIGNMENT", "GPU_FRONT_END", "FUSED_ATTENTION")}
        patch = H._Undo()
        try:
            cpu_kernel_standin.install(patch)              # kernels -> oracle/, unfused attention, backend's own log-mel
            efficient.REUSE_DECODER_LOGITS = False         # a second projection + filters per token (T.py:871-874)
            efficient.DEFER_ALIGNMENT = False              # one synchronous alignment per segment (T.py:544-557)
            model_cpu = H.load_base("cpu")
            all_threads = torch.get_num_threads()
            runs, worst = [], NO_GAPS
            # token-by-token decoding is a chain of small GEMVs: all cores of the box are not the fastest setting, so
            # the baseline is taken at the better of two thread counts (both reported)
            for k, threads in enumerate((min(16, all_threads), all_threads)):
                torch.set_num_threads(threads)
                set_script(Script([window]))
                t0 = time.perf_counter()
                try:
                    r = wt.transcribe(model_cpu, clips[k], language="en", fp16=False)
                finally:
                    set_script(None)
                    torch.set_num_threads(all_threads)
                runs.append({"threads": threads, "seconds_per_clip": round(time.perf_counter() - t0, 2)})
                worst = merge_gaps(worst, word_gaps(words_of(r), words_of(singles[k]), "GPU vs CPU path"))
                if threads == all_threads:
                    break
            # one RAGGED clip as well (its own transcript), at the faster thread count
            best = min(runs, key=lambda x: x["seconds_per_clip"])
            audios, wins, _ = ragged_jobs(B, 100 + B)
            torch.set_num_threads(best["threads"])
            set_script(Script(wins[1]))
            try:
                r = wt.transcribe(model_cpu, audios[1], language="en", fp16=False)
            finally:
                set_script(None)
                torch.set_num_threads(all_threads)
            gpu_same = None
        finally:
            patch.undo()
            for k, v in saved.items():
                setattr(efficient, k, v)
        gpu_same = one(audios[1], wins[1])
        ragged_gap = word_gaps(words_of(gpu_same), words_of(r), "GPU vs CPU path, ragged clip")
        out["cpu_baseline"] = {"value": round(30.0 / best["seconds_per_clip"], 2), "unit": "audio-seconds/s", "cores": best["threads"],
                               "kind": "port", "runs": runs,
                               "sample": f"one 30 s clip per thread setting (the faster one is the baseline), one stream: the same "
                                         f"whisper-base on the CPU, unfused attention with per-token QK capture, second projection "
                                         f"+ logit filters per token, one alignment per segment through oracle/"}
        out["parity_vs_cpu_reference_path"] = gaps_report(worst, {"clips": len(runs), "bars": bars})
        # (a ragged clip as well, reported on its own: the CPU's and the GPU's fp32 GEMMs round differently, and on a
        #  repeated token a random-init model's flat attention leaves the DTW near-ties -- see the note above word_gaps)
        out["parity_vs_cpu_reference_path"]["ragged_clip"] = gaps_report(ragged_gap, {"seconds": round(audios[1].numel() / 16000.0, 2)})
        out["parity_vs_cpu_reference_path"]["ok"] = parity_flag(
            worst[0] <= 0.02 + 1e-9 and worst[1] <= 1e-4 and worst[2] <= 2e-4 and ragged_gap[1] <= 1e-4 and ragged_gap[2] <= 2e-4,
            "default_strategy vs the CPU reference path", out["parity_vs_cpu_reference_path"])
        out["speedup_vs_cpu"] = {k: round(out[k]["audio_s_per_s"] / out["cpu_baseline"]["value"], 1)
                                 for k in ("1_stream", f"{B}_streams", f"{4 * B}_streams", f"ragged_{B}_streams", f"ragged_{4 * B}_streams")}
        emit(out)
    out["parity_failures"] = list(PARITY_FAILURES)           # [] = every parity check of this leg held
    emit(out)
    return out


def role_recordings(args):
    """N > 1 only (BASELINE configs[3] / north_star's "long-audio segment batches shard across the GPUs of one node with RCCL
    broadcast of weights and gather of word-timestamp results"): 32 ragged recordings PER RANK (weak scaling) through
    sharding.transcribe_recordings -- recordings dealt to the ranks largest-first, no data-path collective, every rank steps
    ITS recordings through the decoder together (streams=32), weights broadcast from rank 0, result dictionaries gathered
    to rank 0.  One child process per rank, its own process group (the kernel leg's is gone by now).  --dry-run: gloo, the
    oracle-backed kernel stand-ins, the tiny model, 2 recordings per rank -- the plumbing, not a number."""
    emit = make_emitter(args.out)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    dry = args.dry_run
    import torch.distributed as dist
    from datetime import timedelta
    addr, port = os.environ.get("MASTER_ADDR", "127.0.0.1"), int(os.environ.get("MASTER_PORT", "29533"))
    if os.environ.get("TORCHELASTIC_USE_AGENT_STORE") == "True":
        # under torch.distributed.run the launcher's agent hosts the store: a second group of the same job joins it as a
        # client under its own key prefix (the kernel leg's group left its rendezvous keys behind)
        store = dist.PrefixStore("wt_recordings", dist.TCPStore(addr, port, world, is_master=False, timeout=timedelta(seconds=300)))
    else:
        store = dist.TCPStore(addr, port + 1, world, is_master=(rank == 0), timeout=timedelta(seconds=300))
    import many_helper as H
    import whisper_double as W
    from whisper_double.decoding import Script, set_row_scripts, set_script
    W.install()
    if dry:
        import cpu_kernel_standin
        from test_streams_host import install_streams_standin
        patch = H._Patch()
        cpu_kernel_standin.install(patch)
        install_streams_standin(patch)
        dev = torch.device("cpu")
        dist.init_process_group("gloo", store=store, rank=rank, world_size=world)
        model = H.load_tiny("cpu")
        per_rank, n_streams = 2, 2
    else:
        torch.cuda.set_device(local_rank)
        dev = torch.device("cuda", local_rank)
        dist.init_process_group("nccl", store=store, rank=rank, world_size=world, device_id=dev)
        model = H.load_base(dev)
        per_rank, n_streams = args.e2e_streams, args.e2e_streams
    import whisper_timestamped as wt
    from whisper_timestamped import streams, words
    from whisper_timestamped.sharding import transcribe_recordings
    words.RAW_CONFIDENCE = True
    if rank != 0:                                          # rank 0 holds the truth: the others start from garbage and receive it
        with torch.no_grad():
            for p_ in model.parameters():
                p_.add_(1.0)
    TS0, EOT = 50364, 50257
    g = torch.Generator().manual_seed(7)
    clips = [(0.05 * torch.randn(30 * 16000, generator=g)).float() for _ in range(4)]
    rs = np.random.RandomState(4242)
    audios, wins, secs = [], [], []
    for k in range(per_rank * world):                      # every rank builds the same list; a rank only decodes its own
        sec = float(rs.uniform(5.0, 30.0 if not dry else 8.0))
        audios.append(clips[k % len(clips)][:int(sec * 16000)].clone())
        wins.append([ragged_window(rs, int(sec * 50), TS0, EOT, *((40, 160) if not dry else (8, 16)))])
        secs.append(sec)

    def on_batch(indices):
        scripts = [Script(wins[i]) for i in indices]

        def on_group(rows):
            for r in rows:
                scripts[r].begin_window()
            set_row_scripts([scripts[r] for r in rows])
        streams.ON_GROUP_DECODE = on_group
    sync = (lambda: None) if dry else torch.cuda.synchronize
    opts = dict(language="en", fp16=False)
    try:
        if not dry:                                        # warm-up: allocations, GEMM plans, the communicator
            transcribe_recordings(model, audios, dist=dist, broadcast_weights=True, streams=n_streams, on_batch=on_batch, **opts)
        dist.barrier()
        sync()
        t0 = time.perf_counter()
        res = transcribe_recordings(model, audios, dist=dist, broadcast_weights=dry, streams=n_streams, on_batch=on_batch, **opts)
        sync()
        mine = time.perf_counter() - t0
        dist.barrier()
        el = time.perf_counter() - t0
    finally:
        streams.ON_GROUP_DECODE = None
        set_row_scripts(None)
    every = [None] * world
    dist.all_gather_object(every, round(mine, 4))
    if rank == 0:
        assert len(res) == len(audios) and all(len(r["segments"]) > 0 for r in res)
        worst = NO_GAPS
        picks = sorted({0, len(audios) // 2, len(audios) - 1})
        for k in picks:                                    # recordings other ranks decoded, against one stream here
            set_script(Script(wins[k]))
            try:
                alone = wt.transcribe(model, audios[k], **opts)
            finally:
                set_script(None)
            worst = merge_gaps(worst, word_gaps(words_of(res[k]), words_of(alone), f"recording {k} (another rank) vs one stream"))
        parity_flag(gaps_ok_between_batch_sizes(worst), "transcribe_recordings", gaps_report(worst))
        emit({"parity_failures": list(PARITY_FAILURES), "what": "sharding.transcribe_recordings: ragged recordings (U[5, 30] s, own transcripts) dealt to the ranks, "
                      f"{n_streams} decoder streams per rank, weights broadcast from rank 0 (every other rank started from "
                      "perturbed weights), result dictionaries gathered to rank 0",
              "ranks": world, "recordings": len(audios), "recordings_per_rank": per_rank, "audio_seconds": round(sum(secs), 1),
              "seconds": round(el, 3), "audio_s_per_s": round(sum(secs) / el, 1), "scaling": "weak",
              "per_rank_seconds": every, "backend": "gloo (dry run)" if dry else "rccl",
              "parity_vs_1_stream_on_rank_0": gaps_report(worst, {"recordings_compared": picks})})
    dist.barrier()
    dist.destroy_process_group()


def role_e2e(args):
    emit = make_emitter(args.out)
    if args.inject_fault == "e2e_" + args.leg:
        emit({"marker": "about to abort"})
        os.abort()
    if args.leg == "recordings":
        return role_recordings(args)
    torch.cuda.set_device(0)
    if args.leg == "efficient":
        run_efficient_leg(args, emit)
    else:
        run_e2e(torch.device("cuda", 0), args, args.leg, emit)


def run_child(role, extra, timeout_s, env=None):
    """One measuring leg in a child process of this script.  Returns (result dict or None, error string or None):
    the result is whatever the child published before it ended, however it ended."""
    import subprocess
    import tempfile
    fd, path = tempfile.mkstemp(prefix=f"wt_bench_{role}_", suffix=".json")
    os.close(fd)
    os.unlink(path)
    cmd = [sys.executable, os.path.abspath(__file__)] + sys.argv[1:] + ["--role", role, "--out", path] + list(extra)
    err = None
    # its own session: a leg that has to be stopped takes its own children (worker processes) with it
    proc = subprocess.Popen(cmd, stdout=sys.stderr, env=env, start_new_session=True)      # children never write to stdout
    try:
        rc = proc.wait(timeout=timeout_s)
        if rc < 0:
            err = f"signal {-rc}"
        elif rc != 0:
            err = f"exit {rc}"
    except subprocess.TimeoutExpired:
        err = f"timeout after {timeout_s} s"
        import signal
        try:
            os.killpg(proc.pid, signal.SIGKILL)
        except OSError:
            pass
        proc.wait()
    res = None
    if os.path.exists(path):
        try:
            res = json.load(open(path))
        except Exception:                       # noqa: BLE001
            res = None
        os.unlink(path)
    return res, err


def self_launch(args):
    """`python bench.py --gpus N` outside a launcher: re-exec under torch.distributed.run, one rank per GPU."""
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    if not args.dry_run and torch.cuda.device_count() < args.gpus:
        raise SystemExit(f"bench.py --gpus {args.gpus}: only {torch.cuda.device_count()} GPU(s) visible (one rank per GPU: RCCL "
                         f"refuses two ranks on one device)")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    sys.stdout.flush()
    os.execv(sys.executable, cmd)


def orchestrate(args):
    """The process the driver starts (one per rank).  It never touches the GPU itself: the kernel-level measurement,
    the CPU baselines and each transcribe()-level leg run in child processes that publish their results as they go,
    so a GPU fault in any leg costs that leg (reported as {"error": ...}), not the line.  stdout carries exactly ONE
    JSON line (rank 0), printed when every leg has ended; the headline is also logged to stderr as soon as the
    kernel leg has produced it."""
    world_env = os.environ.get("WORLD_SIZE")
    if args.gpus > 1 and world_env is None:
        self_launch(args)                                  # does not return
    rank = int(os.environ.get("RANK", "0"))
    world = int(world_env or "1")
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    inject = ["--inject-fault", ""]                        # a retried leg is not aborted again
    out, err = run_child("kernel", [], 420)
    attempts = 1
    if (out is None or err) and world == 1:
        # A leg that died is run once more and the line says so: the number is reported, the fault is not hidden.
        first = {"error": err, "published_before_the_fault": sorted(out) if out else None}
        out2, err2 = run_child("kernel", inject, 420)
        attempts = 2
        if out2 is not None and (out is None or not err2):
            out, err = out2, err2
        out = out or {}
        out["kernel_leg_first_attempt"] = first
    multi = None
    if world > 1 and args.e2e != "off":
        # BASELINE configs[3] on N GPUs: recordings across the ranks, decoder streams within a rank (every rank runs its child)
        multi, merr = run_child("e2e", ["--leg", "recordings"], 300)
        multi = multi or {}
        if merr:
            multi["error"] = merr
    if rank != 0:
        sys.exit(1 if err else 0)
    if world > 1 and out is not None:
        out["cpu_baseline"] = "N=1 line only"
        if multi is not None:
            out["transcribe_recordings"] = multi
    if out is None:
        out = {"metric": "audio-seconds aligned/sec (whole node), whisper-base 30s chunks", "value": None, "unit": "audio-seconds/s",
               "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "higher_is_better": True}
    if err:
        out["kernel_leg_error"] = err
    out["kernel_leg_attempts"] = attempts
    # A recurrence of a GPU fault must not read as a clean pass: whatever the retried number is, the line says at its
    # top level that an attempt ended on a signal / non-zero exit.
    out["faulted"] = bool(attempts > 1 or err)
    print("[bench] kernel-level headline: " + json.dumps({k: out.get(k) for k in ("value", "unit", "ms_per_step", "roofline")}),
          file=sys.stderr, flush=True)
    if world == 1 and not args.dry_run and args.workload == "kfull" and args.other_configs == "on":
        # the other single-GPU BASELINE configurations, each as its own kernel-level leg (same child role, same timed
        # region, its own in-leg parity check): kreal = the reference's default per-segment call shape, kfull256 /
        # largev3_fp16 = BASELINE configs[4] shapes at N = 1 (256 chunks; large-v3 heads / mels / vocabulary, fp16 rows)
        others = {}
        for wl in ("kreal", "kfull256", "largev3_fp16"):
            leg, lerr = run_child("kernel", ["--workload", wl, "--min-seconds", "0.5", "--secondary"], 240)
            leg = leg or {}
            keep = {k: leg.get(k) for k in ("value", "unit", "ms_per_step", "timing", "single_batch_in_flight", "roofline",
                                            "stages", "parity_in_leg") if k in leg}
            if "config" in leg:
                keep["workload"] = leg["config"]["workload"]
                keep["alignment_entry"] = leg["config"]["alignment_entry"]
                keep["units_per_step"] = leg["config"].get("units_per_step")
            if lerr:
                keep["error"] = lerr
            others[wl] = keep
        out["other_configs"] = others
    if world == 1 and not args.dry_run:
        fixed_shape = not WORKLOADS[args.workload].get("units_per_chunk")
        if not args.no_cpu_baseline and fixed_shape:       # rank 0 at N=1 only (fixed-shape workloads)
            cpu, cerr = run_child("cpu", [], 300)
            out.update(cpu or {})
            if cerr:
                out["cpu_baseline_error"] = cerr
        if args.e2e == "on" or (args.e2e == "auto" and args.workload == "kfull" and args.overlap == "none" and not args.graph):
            e2e, e1 = run_child("e2e", ["--leg", "fp32"], 420)
            e2e = e2e or {}
            if e1:
                e2e["error"] = e1
            half, e2 = run_child("e2e", ["--leg", "fp16"], 420)
            half = half or {}
            half.pop("marker", None)
            e2e.update(half)
            if e2:
                e2e["fp16_legs_error"] = e2
            eff, e3 = run_child("e2e", ["--leg", "efficient"], 600)
            eff = eff or {}
            eff.pop("marker", None)
            if e3:
                eff["error"] = e3
            e2e["default_strategy"] = eff
            out["e2e_parity_failures"] = eff.get("parity_failures")
            # BASELINE configs[2]: whisper-small shapes through the same second pass (what the reference's beam-search
            # path re-runs, teacher forced: transcribe.py:1197-1262), 32 chunks per launch set
            small, e4 = run_child("e2e", ["--leg", "fp32", "--e2e-model", "small", "--no-cpu-baseline", "--e2e-steps", "3"], 300)
            small = small or {}
            if e4:
                small["error"] = e4
            half_s, e5 = run_child("e2e", ["--leg", "fp16", "--e2e-model", "small", "--e2e-steps", "3"], 300)
            if half_s:
                half_s.pop("marker", None)
                small.update(half_s)
            if e5:
                small["fp16_legs_error"] = e5
            e2e["whisper_small_shapes"] = small
            out["e2e"] = e2e
            # the second half of BASELINE.json's metric ("...; max |dt_word| vs ref"), from the legs that compare words with
            # the reference-shaped CPU path: the teacher-forced second pass and the default strategy
            try:
                dts = [(e2e.get("parity_vs_cpu_reference_path") or {}).get("max_abs_dt_word_s"),
                       (eff.get("parity_vs_cpu_reference_path") or {}).get("max_abs_dt_word_s")]
                dts = [float(x) for x in dts if x is not None]
                out["max_abs_dt_word_vs_ref_s"] = max(dts) if dts else None
            except Exception:                              # noqa: BLE001 -- a summary key must never cost the line
                out["max_abs_dt_word_vs_ref_s"] = None
    print(json.dumps(out, default=_json_scalar), flush=True)
    if out.get("value") is None:
        sys.exit(1)


def cap_threads_per_rank():
    """N ranks on one host: each gets its share of the cores for torch's intra-op pool (and its children inherit it) --
    eight ranks with every core each is what made the default-strategy CPU leg 12x slower at 128 threads than at 16."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1:
        n = max(1, (os.cpu_count() or world) // world)
        torch.set_num_threads(n)
        os.environ["OMP_NUM_THREADS"] = str(n)
    return torch.get_num_threads()


def main():
    args = parse_args()
    cap_threads_per_rank()
    if args.role == "orchestrate":
        return orchestrate(args)
    # children: stdout belongs to the parent's single JSON line -- everything libraries print goes to stderr
    sys.stdout.flush()
    os.dup2(2, 1)
    {"kernel": role_kernel, "cpu": role_cpu, "e2e": role_e2e}[args.role](args)


if __name__ == "__main__":
    main()
