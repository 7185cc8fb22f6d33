"""Synthetic workloads of the hot path (SURVEY.md 8(d) sets K-full / K-real and the BASELINE configs' shapes), shared
by bench.py's kernel-level leg, the GPU tests and the fenced-buffer runner; and the in-leg parity check that holds a few
units of a timed batch against the oracle.  A workload is a dictionary of the tensors it was drawn as plus
``w["batch"]``: the product's ``whisper_timestamped.pipeline.ChunkBatch`` over those tensors (the thing that is launched).
Test / measurement infrastructure: nothing in the package imports this."""
import numpy as np
import torch

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
    return bind_batch(w, cfg.get('align', 'split'))


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
    return bind_batch(dict(cfg=cfg, qk=qk, logits=logits, tokens=tokens, pcm=pcm, fb=mel_filters(dev, cfg["n_mels"]), descs=descs,
                n_valid=torch.from_numpy(n_valid).to(dev), parity_units=sample, qk_f32_sample=None,
                unit_chunk=[raw[i]["chunk"] for i in order], unit_row0=[raw[i]["row0"] for i in order],
                unit_logit_row0=[raw[i]["logit_row0"] for i in order],
                descs_dev=_lib.descs_to_device(descs, dev), head_idx=torch.arange(A, dtype=torch.int32, device=dev),
                cost=torch.empty(n_cost, dtype=torch.float32, device=dev),
                **result_buffers(n_jumps, tot_T, dev),
                mel=torch.empty((n, cfg["n_mels"], 3000), dtype=torch.float32, device=dev),
                gmax=torch.empty(n, dtype=torch.float32, device=dev), pad=torch.empty(n, dtype=torch.int32, device=dev),
                stairs=stairs), cfg.get('align', 'fused'))

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



def result_buffers(n_jumps, n_logprob, dev):
    """pipeline.result_record as the workload dictionary's entries (one device record: jumps | log-probabilities)."""
    from whisper_timestamped.pipeline import result_record
    rec, host = result_record(n_jumps, n_logprob, dev)
    return dict(result=rec, jumps=rec[:n_jumps], logprob=rec[n_jumps:].view(torch.float32), host_result=host,
                host_jumps=host[:n_jumps], host_logprob=host[n_jumps:].view(torch.float32))


def bind_batch(w, align=None):
    """(Re)build w["batch"] from the dictionary's tensors (after a caller swapped buffers, e.g. for fenced copies)."""
    from whisper_timestamped.pipeline import ChunkBatch
    if align is not None:
        w["align"] = align
    n_jumps = w["jumps"].numel()
    w["batch"] = ChunkBatch(w["pcm"], w["n_valid"], w["fb"], w["qk"], w["descs"], w["descs_dev"], w["head_idx"], w["logits"],
                            w["tokens"], w["mel"], w["gmax"], w["pad"], w["cost"], w["result"], w["host_result"], n_jumps,
                            fused_small_units=(w.get("align") == "fused"))
    return w


def twin(w):
    """A second buffer set over the same inputs (its own outputs and record), as a workload dictionary."""
    c = dict(w)
    b = w["batch"].twin()
    c.update(batch=b, cost=b.cost, mel=b.mel, gmax=b.gmax, pad=b.pad, result=b.result, jumps=b.jumps, logprob=b.logprob,
             host_result=b.host_result, host_jumps=b.host_jumps, host_logprob=b.host_logprob)
    return c


_SERIAL = {}


def run_step(w, stage_set=None):
    """One pass of the hot path over the workload on the CURRENT stream, stages in order (or on the streams of `stage_set`)."""
    from whisper_timestamped.pipeline import HotPathPipeline, StageSet
    if "batch" not in w:
        bind_batch(w)
    s = stage_set or StageSet.on_current_stream(w["batch"].device)
    HotPathPipeline._stages(s, w["batch"], w["batch"])
    s.join(w["batch"].fetch, mark=w["batch"].copied)
