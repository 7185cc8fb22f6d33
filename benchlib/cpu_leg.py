"""cpu_baseline: the oracle (CPU restatement of the reference path) timed on the host cores of the GPU box.  The only
place besides the parity checks where bench.py runs anything under oracle/ -- as the baseline, never as the product."""
import os
import sys
import time

import numpy as np
import torch

from .common import log, make_emitter


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



def role_cpu(args):
    """cpu_baseline: the oracle on the host cores over a bounded sample of the same workload (the sample is drawn on
    the GPU with the kernel leg's generator, then moved to the host)."""
    emit = make_emitter(args.out)
    dev = torch.device("cuda", 0)
    import workloads as WL
    cfg = WL.WORKLOADS[args.workload]
    w = WL.make_workload(dev, cfg, seed=1234)
    out = {"cpu_baseline": cpu_baseline(w["cfg"], w)}
    emit(out)
    out["cpu_baseline_1thread"] = cpu_baseline(w["cfg"], w, budget_s=8.0, threads=1)
    emit(out)

