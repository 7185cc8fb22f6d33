#!/usr/bin/env python3
"""Runs the kernels of libwtalign.so on buffers whose neighbours are UNMAPPED pages (tests/guard/guard_alloc.cpp).

Test infrastructure.  Every input and output of a call is placed so that it ENDS at the last mapped byte of its
mapping (mode "end": an overrun by one element is a GPU memory access fault, i.e. this process dies) or STARTS at
the first mapped byte (mode "start": the same for an underrun), and the results are compared bit for bit with the
same call on ordinary torch tensors.  tests/test_gpu_guard.py runs each case in a child process (a fault must not
take the test session down) and asserts exit code 0.

    python tests/guard/run_guarded.py CASE MODE        CASE in CASES, MODE in {end, start}
"""
import ctypes
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
for p in (ROOT, os.path.join(ROOT, "whisper-timestamped_amd"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

_G = None


def guard_lib():
    global _G
    if _G is None:
        path = os.path.join(HERE, "libwtguard.so")
        if not os.path.exists(path):
            raise ImportError(f"{path} is missing: __graft_entry__.build() compiles it")
        _G = ctypes.CDLL(path)
        _G.wt_guard_alloc.restype = ctypes.c_void_p
        _G.wt_guard_alloc.argtypes = [ctypes.c_size_t, ctypes.c_int, ctypes.POINTER(ctypes.c_void_p)]
        _G.wt_guard_free.argtypes = [ctypes.c_void_p]
        _G.wt_guard_error.restype = ctypes.c_char_p
    return _G


class _Raw:
    def __init__(self, ptr, shape, typestr):
        self.__cuda_array_interface__ = {"shape": tuple(shape), "typestr": typestr, "data": (ptr, False), "version": 2}


_TYPESTR = {torch.float32: "<f4", torch.float16: "<f2", torch.int32: "<i4", torch.uint8: "|u1", torch.float64: "<f8",
            torch.int64: "<i8"}
_KEEP = []     # tickets (mappings stay alive for the life of the process)


def guarded(t: torch.Tensor, mode: str) -> torch.Tensor:
    """A copy of `t` (contiguous) whose storage is fenced by unmapped pages; same shape / dtype."""
    t = t.contiguous()
    nbytes = t.numel() * t.element_size()
    ticket = ctypes.c_void_p()
    ptr = guard_lib().wt_guard_alloc(max(nbytes, 1), 1 if mode == "end" else 0, ctypes.byref(ticket))
    if not ptr:
        raise RuntimeError("wt_guard_alloc: " + guard_lib().wt_guard_error().decode())
    _KEEP.append(ticket)
    if t.numel() == 0:
        return t
    g = torch.as_tensor(_Raw(ptr, t.shape, _TYPESTR[t.dtype]), device=t.device)
    assert g.data_ptr() == ptr
    g.copy_(t)
    return g


def guard_workload(w, mode):
    import workloads as WL
    out = dict(w)
    out.pop("batch", None)
    for k in ("qk", "logits", "tokens", "pcm", "n_valid", "fb", "descs_dev", "head_idx", "cost", "mel", "gmax", "pad", "result"):
        out[k] = guarded(w[k], mode)
    n_jumps = w["jumps"].numel()
    out["jumps"] = out["result"][:n_jumps]
    out["logprob"] = out["result"][n_jumps:].view(torch.float32)
    return WL.bind_batch(out)


def case_step(workload, mode, n_chunks=None):
    """One bench step (log-mel, padding, cost, DTW, log-prob gather) with every buffer fenced."""
    import workloads as WL
    cfg = dict(WL.WORKLOADS[workload])
    if n_chunks:
        cfg["n_chunks"] = n_chunks
    dev = torch.device("cuda", 0)
    w = WL.make_workload(dev, cfg, seed=77)
    WL.run_step(w)
    torch.cuda.synchronize()
    g = guard_workload(w, mode)
    for k in ("cost", "mel", "result"):
        g[k].fill_(0)
    WL.run_step(g)
    WL.run_step(g)           # (twice: the second call reuses the library's arenas)
    torch.cuda.synchronize()
    for k in ("jumps", "mel", "pad", "gmax"):
        assert torch.equal(g[k], w[k]), f"{workload}/{mode}: {k} differs between fenced and ordinary buffers"
    # (the log-softmax sums a row as head / 16-byte body / tail split at the row's first 128-byte line: the fenced copy
    # of the logits sits on another alignment, so its sums may round differently in the last place)
    assert torch.allclose(g["logprob"], w["logprob"], rtol=0, atol=2e-5), f"{workload}/{mode}: logprob"
    n_cost = sum(int(d["T"]) * int(d["F"]) for d in w["descs"])
    assert n_cost > 0
    for d in w["descs"]:
        c0, n = int(d["cost_offset"]), int(d["T"]) * int(d["F"])
        assert torch.equal(g["cost"][c0:c0 + n], w["cost"][c0:c0 + n])


def case_odd_units(mode, dtype, flags=1):
    """Cost + DTW (+ path, distance, disfluency) on units of awkward shapes: F not a multiple of 4, tiny, maximal.
    flags = WT_ALIGN_KEEP_COST (small units through the fused kernel) or | WT_ALIGN_NO_FUSED_SMALL_UNITS (all batched)."""
    from whisper_timestamped import _lib as L
    import synth
    dev = "cuda:0"
    shapes = [(1, 0, 1), (2, 0, 3), (3, 10, 14), (5, 1, 8), (9, 100, 245), (17, 275, 523), (11, 3, 258), (64, 0, 1500),
              (65, 219, 1500), (31, 1, 770), (130, 0, 1281), (224, 0, 1500), (255, 7, 1499), (64, 3, 258), (16, 0, 1500),
              (30, 100, 1125), (7, 1, 513), (74, 100, 400), (130, 0, 60), (150, 2, 32)]
    heads = list(range(6))
    qk_list = [synth.synth_qk(300 + k, 6, T, lo=s, hi=e) for k, (T, s, e) in enumerate(shapes)]
    order = L.launch_order([(T, e - s) for T, s, e in shapes])
    descs = L.make_descs(len(shapes))
    off, offs = 0, []
    for q in qk_list:
        offs.append(off)
        off += q.size
    for d, i in zip(descs, order):
        T, s, e = shapes[i]
        q = qk_list[i]
        d["qk_offset"], d["head_stride"], d["row_stride"] = offs[i], q.shape[1] * q.shape[2], q.shape[2]
        d["T"], d["F"], d["start_token"], d["pad_from"] = T, e - s, s, (-1 if i % 3 else max((e - s) // 2, 1))
    n_cost, n_jumps, n_path = L.layout_outputs(descs)
    qk = torch.from_numpy(np.concatenate([q.ravel() for q in qk_list])).to(dev).to(dtype)

    def run(fence):
        f = (lambda t: guarded(t, mode)) if fence else (lambda t: t)
        b = dict(qk=f(qk), dd=f(L.descs_to_device(descs, dev)), hi=f(torch.tensor(heads, dtype=torch.int32, device=dev)),
                 cost=f(torch.zeros(n_cost, dtype=torch.float32, device=dev)),
                 jumps=f(torch.zeros(n_jumps, dtype=torch.int32, device=dev)),
                 starts=f(torch.zeros(n_jumps, dtype=torch.int32, device=dev)),
                 pi=f(torch.zeros(n_path, dtype=torch.int32, device=dev)), pj=f(torch.zeros(n_path, dtype=torch.int32, device=dev)),
                 pl=f(torch.zeros(len(shapes), dtype=torch.int32, device=dev)),
                 dist=f(torch.zeros(len(shapes), dtype=torch.float64, device=dev)))
        L.align_batch(b["qk"], descs, b["dd"], b["hi"], b["cost"], b["jumps"], b["pi"], b["pj"], b["pl"], b["dist"], flags=flags)
        st = torch.cuda.current_stream().cuda_stream
        L._check(L.load().wt_disfluency_batch(b["cost"].data_ptr(), b["dd"].data_ptr(), len(shapes), b["jumps"].data_ptr(),
                                              b["starts"].data_ptr(), 0.02, 3.0, st), "wt_disfluency_batch")
        # the other step pattern (no same-frame token moves) where it has a path
        ok = [k for k, d in enumerate(descs) if d["T"] <= d["F"]]
        sub = descs[ok].copy()
        sd = f(L.descs_to_device(sub, dev))
        j2 = f(torch.zeros(n_jumps, dtype=torch.int32, device=dev))
        L.dtw_batch(b["cost"], sub, sd, j2, step_pattern=L.WT_STEP_NO_EMPTY_SUBWORDS)
        torch.cuda.synchronize()
        b["j2"] = j2
        return b

    plain, fenced = run(False), run(True)
    for k in ("jumps", "starts", "pl", "dist", "j2"):
        assert torch.equal(plain[k], fenced[k]), k
    for d in descs:
        c0, n = int(d["cost_offset"]), int(d["T"]) * int(d["F"])
        assert torch.equal(plain["cost"][c0:c0 + n], fenced["cost"][c0:c0 + n])


def case_logprob(mode):
    """Row strides that put the rows on every alignment, fp32 and fp16, masks, the row-index form."""
    from whisper_timestamped import _lib as L
    dev = "cuda:0"
    g = torch.Generator(device=dev).manual_seed(5)
    for V, n, dtype in [(51865, 9, torch.float32), (51866, 5, torch.float16), (51864, 3, torch.float32), (1000, 7, torch.float32),
                        (7, 3, torch.float32), (33, 4, torch.float16), (1, 2, torch.float32)]:
        logits = (torch.randn((n, V), generator=g, device=dev) * 4).to(dtype)
        toks = torch.randint(0, V, (n,), generator=g, device=dev, dtype=torch.int32)
        mask = (torch.rand((n, V), generator=g, device=dev) < 0.3)
        mask[torch.arange(n), toks.long()] = False
        idx = torch.randint(0, n, (11,), generator=g, device=dev, dtype=torch.int32)
        tk2 = torch.randint(0, V, (11,), generator=g, device=dev, dtype=torch.int32)
        want = L.logprob_gather(logits, toks)
        want_m = L.logprob_gather(logits, toks, mask)
        want_r = L.logprob_gather_rows(logits, idx, tk2)
        gl, gt, gm = guarded(logits, mode), guarded(toks, mode), guarded(mask.to(torch.uint8), mode)
        st = torch.cuda.current_stream().cuda_stream
        dt = 0 if dtype == torch.float32 else 1
        for sup, srows, ref in [(0, 0, want), (gm.data_ptr(), n, want_m)]:
            out = guarded(torch.zeros(n, device=dev), mode)
            L._check(L.load().wt_logprob_gather_batch(gl.data_ptr(), dt, V, n, V, gt.data_ptr(), sup, srows, out.data_ptr(), st), "gather")
            torch.cuda.synchronize()
            assert torch.allclose(out, ref, rtol=0, atol=2e-5), (V, n, dtype)     # (alignment-dependent summation order)
        out = guarded(torch.zeros(11, device=dev), mode)
        L._check(L.load().wt_logprob_gather_rows(gl.data_ptr(), dt, V, guarded(idx, mode).data_ptr(), 11, V,
                                                 guarded(tk2, mode).data_ptr(), out.data_ptr(), st), "gather_rows")
        torch.cuda.synchronize()
        assert torch.allclose(out, want_r, rtol=0, atol=2e-5), (V, n, dtype)


def case_digest(mode):
    """wt_logprob_digest_streams: rows of every alignment (row stride n_q * V and V), int32 / strided int64 tokens, a subset
    of the ring blocks in any order, with and without the timestamp slice; fenced rows, tokens, ring index and rings."""
    from whisper_timestamped import _lib as L
    dev = "cuda:0"
    g = torch.Generator(device=dev).manual_seed(15)
    for V, n_q, slice_begin, i64 in [(51865, 3, 50364, True), (51866, 1, 50365, False), (1000, 2, 990, True), (37, 1, 30, False)]:
        n_blocks, rows_cap, gsz = 5, 6, 3
        outs = torch.randn((gsz, n_q, V), generator=g, device=dev) * 4
        toks_full = torch.randint(0, V, (gsz, 4), generator=g, device=dev, dtype=torch.int64 if i64 else torch.int32)
        ring_index = torch.tensor([4, 0, 2], dtype=torch.int32, device=dev)
        aux = [V - 1, 0]
        res = []
        for fence in (False, True):
            f = (lambda t: guarded(t, mode)) if fence else (lambda t: t.clone())
            o, t, ri = f(outs), f(toks_full), f(ring_index)
            digest = f(torch.zeros((n_blocks, rows_cap, L.DIGEST_WORDS), device=dev))
            sl = f(torch.zeros((n_blocks, rows_cap, V - slice_begin), device=dev))
            for step in (0, rows_cap - 1):
                L.logprob_digest_streams(o[:, -1], t[:, -1], ri, digest, sl, step, aux, slice_begin)
            L.logprob_digest_streams(o[:, -1], t[:, -1], ri, digest, None, 1, aux, slice_begin)
            torch.cuda.synchronize()
            res.append((digest.clone(), sl.clone()))
        a, b = res[0][0], res[1][0]
        # (the split of a row into head / 16-byte body / tail follows the row's address: the summation order, hence the last
        #  bits of log-sum-exp, depend on the alignment -- as for wt_logprob_gather_batch; everything else is exact)
        assert torch.allclose(a[..., :3], b[..., :3], rtol=0, atol=2e-5), (V, n_q)
        assert torch.equal(a[..., 1], b[..., 1]) and torch.equal(a[..., 3:].view(torch.int32), b[..., 3:].view(torch.int32)), (V, n_q)
        assert torch.equal(res[0][1], res[1][1]), (V, n_q)
        want = L.logprob_gather(outs[:, -1], toks_full[:, -1].to(torch.int32))
        assert torch.equal(res[0][0][ring_index.long(), 0, 0], want), (V, n_q)          # same rows, same alignment: same bits
        assert torch.equal(res[1][1][ring_index.long(), rows_cap - 1], outs[:, -1, slice_begin:])


def case_logmel(mode):
    """Whole and ragged chunks, lengths that are not multiples of 4 / 160, 80 and 128 mel bins, + the padding detector."""
    from whisper_timestamped import _lib as L
    from whisper_timestamped.audio import mel_filters
    dev = "cuda:0"
    g = torch.Generator(device=dev).manual_seed(6)
    st = torch.cuda.current_stream().cuda_stream
    for B, N, n_mels, n_frames, ragged in [(3, 480000, 80, 3000, False), (5, 480000, 128, 3000, True), (2, 201, 80, 3000, False),
                                           (4, 16001, 80, 101, False), (3, 33333, 80, 3000, True), (7, 160 * 37, 128, 37, False)]:
        pcm = torch.randn((B, N), generator=g, device=dev) * 0.1
        fb = mel_filters(dev, n_mels)
        nv = None
        if ragged:
            nv = torch.randint(201, N + 1, (B,), generator=g, device=dev, dtype=torch.int32)
            nv[0] = N
        want, want_max = L.logmel(pcm, fb, nv, n_frames)
        want_pad = L.find_start_padding(want)
        gp, gf = guarded(pcm, mode), guarded(fb, mode)
        gn = guarded(nv, mode) if nv is not None else None
        mel = guarded(torch.zeros((B, n_mels, n_frames), device=dev), mode)
        gmax = guarded(torch.zeros(B, device=dev), mode)
        pad = guarded(torch.zeros(B, dtype=torch.int32, device=dev), mode)
        pad2 = guarded(torch.full((B,), 77, dtype=torch.int32, device=dev), mode)
        for _ in range(2):
            L._check(L.load().wt_logmel_batch(gp.data_ptr(), B, N, L._ptr(gn), gf.data_ptr(), n_mels, n_frames, mel.data_ptr(),
                                              gmax.data_ptr(), st), "wt_logmel_batch")
            L._check(L.load().wt_find_start_padding_batch(mel.data_ptr(), B, n_mels, n_frames, pad.data_ptr(), st), "padding")
        torch.cuda.synchronize()
        assert torch.equal(mel, want) and torch.equal(gmax, want_max) and torch.equal(pad, want_pad), (B, N, n_mels, n_frames)
        mel.zero_()
        for _ in range(2):                 # the same with the detector folded into the finalising pass
            L._check(L.load().wt_logmel_pad_batch(gp.data_ptr(), B, N, L._ptr(gn), gf.data_ptr(), n_mels, n_frames, mel.data_ptr(),
                                                  gmax.data_ptr(), pad2.data_ptr(), st), "wt_logmel_pad_batch")
        torch.cuda.synchronize()
        assert torch.equal(mel, want) and torch.equal(gmax, want_max) and torch.equal(pad2, want_pad), (B, N, n_mels, n_frames, pad2, want_pad)


def case_capture(mode):
    """wt_capture_rows, wt_qk_rows and wt_qk_rows_batch (f32 and f16 sources and rings)."""
    from whisper_timestamped import _lib as L
    dev = "cuda:0"
    g = torch.Generator(device=dev).manual_seed(9)
    st = torch.cuda.current_stream().cuda_stream
    lib = L.load()
    D, H, n_ctx = 512, 8, 1500
    for src_dt, ring_dt in [(torch.float32, torch.float32), (torch.float16, torch.float16), (torch.float32, torch.float16)]:
        code = {torch.float32: 0, torch.float16: 1}
        # observed rows
        n_q, rows = 5, 12
        qk = torch.randn((H, n_q, n_ctx), generator=g, device=dev).to(src_dt)
        heads = torch.tensor([1, 4, 7], dtype=torch.int32, device=dev)
        slots = torch.tensor([0, 1, 2], dtype=torch.int32, device=dev)
        res = []
        for fence in (False, True):
            f = (lambda t: guarded(t, mode)) if fence else (lambda t: t)
            ring = f(torch.zeros((3, rows, n_ctx), device=dev, dtype=ring_dt))
            a = (f(qk), f(heads), f(slots))
            L._check(lib.wt_capture_rows(a[0].data_ptr(), code[src_dt], H, n_q, n_ctx, a[1].data_ptr(), a[2].data_ptr(), 3,
                                         ring.data_ptr(), code[ring_dt], rows, rows - 1, st), "wt_capture_rows")
            # rows from the projections
            q = torch.randn((4, D), generator=torch.Generator(device=dev).manual_seed(1), device=dev).to(src_dt)
            k = torch.randn((n_ctx, D), generator=torch.Generator(device=dev).manual_seed(2), device=dev).to(src_dt)
            b = (f(q), f(k))
            L._check(lib.wt_qk_rows(b[0].data_ptr(), b[1].data_ptr(), code[src_dt], 4, n_ctx, D, 64, 64 ** -0.25, a[1].data_ptr(),
                                    a[2].data_ptr(), 3, ring.data_ptr(), code[ring_dt], rows, 2, st), "wt_qk_rows")
            # batched form: 3 windows, 2 layers
            B, n_qb = 3, 7
            ql = [f(torch.randn((B, n_qb, D), generator=torch.Generator(device=dev).manual_seed(10 + l), device=dev).to(src_dt)) for l in range(2)]
            kl = [f(torch.randn((B, n_ctx, D), generator=torch.Generator(device=dev).manual_seed(20 + l), device=dev).to(src_dt)) for l in range(2)]
            ringb = f(torch.zeros((B, 4, 9, n_ctx), device=dev, dtype=ring_dt))
            sl = f(torch.tensor([0, 0, 1, 1], dtype=torch.int32, device=dev))
            sh = f(torch.tensor([2, 5, 0, 7], dtype=torch.int32, device=dev))
            ss = f(torch.tensor([0, 1, 2, 3], dtype=torch.int32, device=dev))
            rb = f(torch.tensor([0, 1, 2], dtype=torch.int32, device=dev))
            re = f(torch.tensor([7, 6, 7], dtype=torch.int32, device=dev))
            L.qk_rows_batch(ql, kl, sl, sh, ss, ringb, rb, re, ring_row0=1)
            torch.cuda.synchronize()
            res.append((ring, ringb))
        assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1]), (src_dt, ring_dt)


CASES = {
    "step_kfull": lambda m: case_step("kfull", m),
    "step_kreal": lambda m: case_step("kreal", m),
    "step_largev3_fp16": lambda m: case_step("largev3_fp16", m, n_chunks=8),
    "odd_units_f32": lambda m: case_odd_units(m, torch.float32),
    "odd_units_f16": lambda m: case_odd_units(m, torch.float16),
    "odd_units_f32_batched_only": lambda m: case_odd_units(m, torch.float32, flags=3),
    "odd_units_f16_batched_only": lambda m: case_odd_units(m, torch.float16, flags=3),
    "odd_units_f32_rows_per_class": lambda m: case_odd_units(m, torch.float32, flags=5),
    "odd_units_f16_rows_per_class": lambda m: case_odd_units(m, torch.float16, flags=5),
    "logprob": case_logprob,
    "digest": case_digest,
    "logmel": case_logmel,
    "capture": case_capture,
}

if __name__ == "__main__":
    name, mode = sys.argv[1], sys.argv[2]
    assert mode in ("end", "start")
    torch.cuda.set_device(0)
    CASES[name](mode)
    print(f"guarded {name}/{mode}: ok")
