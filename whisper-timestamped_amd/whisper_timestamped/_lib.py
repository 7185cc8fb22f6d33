"""ctypes binding of libwtalign.so (C ABI: include/wtalign.h).

The HIP library is the product path: there is NO CPU fallback.  Importing this
module without the built library raises; calling into it without a GPU raises.
"""
from __future__ import annotations

import ctypes
import os
import threading

import numpy as np
import torch

_PKG_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# (WT_LIBWTALIGN: another build of the same library, for A/B measurements of kernel variants -- tools/ab_cost.py)
LIB_PATH = os.environ.get("WT_LIBWTALIGN") or os.path.join(_PKG_ROOT, "libwtalign.so")

WT_DTYPE_F32, WT_DTYPE_F16 = 0, 1
WT_MAX_TOKENS, WT_MAX_FRAMES = 256, 1792
N_AUDIO_CTX = 1500

# mirrors `struct wt_seg_desc` (64 bytes)
SEG_DTYPE = np.dtype([
    ("qk_offset", "<i8"), ("head_stride", "<i8"), ("row_stride", "<i8"), ("cost_offset", "<i8"),
    ("jumps_offset", "<i8"), ("path_offset", "<i8"), ("T", "<i4"), ("F", "<i4"), ("start_token", "<i4"),
    ("pad_from", "<i4"),
], align=True)
assert SEG_DTYPE.itemsize == 64

EXPORTS = ["wt_version", "wt_last_error", "wt_shutdown", "wt_cost_batch", "wt_dtw_batch", "wt_align_batch",
           "wt_find_start_padding_batch", "wt_logprob_gather_batch", "wt_logmel_batch", "wt_capture_rows", "wt_qk_rows",
           "wt_disfluency_batch", "wt_qk_rows_batch", "wt_logprob_gather_rows", "wt_dtw_batch_pattern", "wt_align_batch_v3",
           "wt_release_stream", "wt_qk_rows_streams", "wt_logmel_pad_batch", "wt_logprob_digest_streams"]
WT_STEP_SYMMETRIC1, WT_STEP_NO_EMPTY_SUBWORDS = 0, 1
ABI_VERSION = 5
WT_ALIGN_KEEP_COST, WT_ALIGN_NO_FUSED_SMALL_UNITS, WT_ALIGN_ROWS_PER_CLASS = 1, 2, 4


class WtError(RuntimeError):
    pass


_lib = None


def load():
    """Load libwtalign.so (raises if it has not been built: `python __graft_entry__.py`)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(f"{LIB_PATH} is missing: build it with `make -C {_PKG_ROOT}/csrc` "
                          "(or __graft_entry__.build()); there is no CPU fallback")
    L = ctypes.CDLL(LIB_PATH)
    vp, i32, i64, f32 = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_float
    L.wt_version.restype = i32
    L.wt_last_error.restype = ctypes.c_char_p
    L.wt_shutdown.restype = i32
    L.wt_cost_batch.argtypes = [vp, i32, vp, vp, i32, vp, i32, i32, f32, vp, vp]
    L.wt_dtw_batch.argtypes = [vp, vp, vp, i32, vp, vp, vp, vp, vp, vp]
    L.wt_align_batch.argtypes = [vp, i32, vp, vp, i32, vp, i32, i32, f32, vp, vp, vp, vp, vp, vp, vp]
    L.wt_align_batch_v3.argtypes = [vp, i32, vp, vp, i32, vp, i32, i32, f32, vp, vp, vp, vp, vp, vp, i32, vp]
    L.wt_find_start_padding_batch.argtypes = [vp, i32, i32, i32, vp, vp]
    L.wt_disfluency_batch.argtypes = [vp, vp, i32, vp, vp, ctypes.c_double, ctypes.c_double, vp]
    L.wt_logprob_gather_batch.argtypes = [vp, i32, i64, i32, i32, vp, vp, i32, vp, vp]
    L.wt_logmel_batch.argtypes = [vp, i32, i64, vp, vp, i32, i32, vp, vp, vp]
    L.wt_capture_rows.argtypes = [vp, i32, i32, i32, i32, vp, vp, i32, vp, i32, i64, i64, vp]
    L.wt_qk_rows.argtypes = [vp, vp, i32, i32, i32, i32, i32, f32, vp, vp, i32, vp, i32, i64, i64, vp]
    L.wt_qk_rows_batch.argtypes = [vp, vp, i32, i32, i32, i32, i64, i64, i32, i32, i32, f32, vp, vp, vp, i32, vp, vp, vp, i32, i64,
                                   i64, i64, vp]
    L.wt_logprob_gather_rows.argtypes = [vp, i32, i64, vp, i32, i32, vp, vp, vp]
    L.wt_dtw_batch_pattern.argtypes = [vp, vp, vp, i32, i32, vp, vp, vp, vp, vp, vp]
    L.wt_release_stream.argtypes = [vp]
    L.wt_logmel_pad_batch.argtypes = [vp, i32, i64, vp, vp, i32, i32, vp, vp, vp, vp]
    L.wt_qk_rows_streams.argtypes = [vp, vp, i32, i32, i32, i32, i64, i64, i32, i32, i32, f32, vp, vp, vp, i32, vp, vp, i32, i64,
                                     i64, i64, vp]
    L.wt_logprob_digest_streams.argtypes = [vp, i64, i32, i32, vp, i32, i64, vp, i64, i64, vp, i32, i32, vp, vp, vp]
    if L.wt_version() != ABI_VERSION:
        raise ImportError(f"{LIB_PATH} exports ABI version {L.wt_version()}, this package needs {ABI_VERSION}: rebuild it "
                          f"(`make -C {_PKG_ROOT}/csrc`)")
    for n in EXPORTS[3:]:
        getattr(L, n).restype = i32
    _lib = L
    return L


def _check(rc: int, what: str):
    if rc != 0:
        msg = load().wt_last_error().decode("utf-8", "replace")
        raise WtError(f"{what} failed (rc={rc}): {msg}")


def release_stream(stream) -> int:
    """Free the scratch arenas the library keeps for a HIP stream (a ``torch.cuda.Stream`` or a raw handle) whose work
    has completed: call it before dropping a stream the kernels were launched on.  Returns the number of arenas freed."""
    handle = stream.cuda_stream if hasattr(stream, "cuda_stream") else int(stream)
    rc = load().wt_release_stream(handle)
    if rc < 0:
        _check(rc, "wt_release_stream")
    return rc


def _stream(device=None) -> int:
    """Current torch stream of `device` (default: the current device)."""
    return torch.cuda.current_stream(device).cuda_stream


class on_device:
    """``with on_device(t) as stream:`` -- the library launches on, and keeps its scratch arenas per, the CURRENT HIP
    device; a tensor that lives on another GPU (model loaded on cuda:1 without torch.cuda.set_device) would
    otherwise get its kernels launched on device 0 with device-1 pointers.  Makes the tensor's device current for the
    call and hands out that device's current stream."""

    def __init__(self, t):
        dev = t.device if isinstance(t, torch.Tensor) else torch.device(t)
        if dev.type != "cuda":
            raise WtError("the alignment kernels only run on the GPU (got a tensor on %s)" % dev)
        self.ctx = torch.cuda.device(dev)
        self.dev = dev

    def __enter__(self):
        self.ctx.__enter__()
        return torch.cuda.current_stream(self.dev).cuda_stream

    def __exit__(self, *exc):
        return self.ctx.__exit__(*exc)


def device_ctx(device):
    """torch.cuda.device(device) for a GPU device, a no-op otherwise."""
    import contextlib
    device = torch.device(device)
    return torch.cuda.device(device) if device.type == "cuda" else contextlib.nullcontext()


# Page-locked staging buffers are expensive to create (hipHostMalloc: milliseconds) and tiny here (token matrices,
# padding indices, gather lists): they are pooled per byte-size bucket and handed back after use.
_PINNED_FREE = {}
_PINNED_LOCK = threading.Lock()


def _pinned_take(nbytes: int) -> torch.Tensor:
    bucket = max(256, 1 << (max(int(nbytes), 1) - 1).bit_length())
    with _PINNED_LOCK:
        free = _PINNED_FREE.setdefault(bucket, [])
        if free:
            return free.pop()
    return torch.empty(bucket, dtype=torch.uint8).pin_memory()


def _pinned_give(buf: torch.Tensor):
    with _PINNED_LOCK:
        _PINNED_FREE.setdefault(buf.numel(), []).append(buf)


class PinnedUpload:
    """Small host array -> device through pooled page-locked memory, asynchronously.  ``release()`` (after the work
    that consumed it has been waited for) hands the staging buffer back to the pool."""

    def __init__(self, array, device):
        src = torch.from_numpy(np.ascontiguousarray(array))
        self.buf = None
        if torch.device(device).type == "cuda":
            nbytes = src.numel() * src.element_size()
            self.buf = _pinned_take(nbytes)
            stage = self.buf[:nbytes].view(src.dtype)
            stage.copy_(src)
            self.dev = stage.to(device, non_blocking=True)
        else:
            self.dev = src

    def release(self):
        if self.buf is not None:
            _pinned_give(self.buf)
            self.buf = None


class HostCopy:
    """Asynchronous device->host copy of a small tensor: queued behind the work already on the stream, read with
    ``wait()`` (pooled page-locked staging; a tensor that already lives on the host is handed back as is)."""

    def __init__(self, t: torch.Tensor):
        self.event = self.buf = None
        if t.is_cuda:
            nbytes = t.numel() * t.element_size()
            self.buf = _pinned_take(nbytes)
            self.host = self.buf[:nbytes].view(t.dtype).view(t.shape)
            with torch.cuda.device(t.device):
                self.host.copy_(t.contiguous(), non_blocking=True)
                self.event = torch.cuda.Event()
                self.event.record(torch.cuda.current_stream(t.device))
        else:
            self.host = t

    def wait(self) -> torch.Tensor:
        if self.event is not None:
            self.event.synchronize()
            self.event = None
            self.host = self.host.clone()          # (bytes) -- the staging buffer goes back to the pool
            _pinned_give(self.buf)
            self.buf = None
        return self.host


def same_device(*tensors):
    devs = {t.device for t in tensors if isinstance(t, torch.Tensor)}
    if len(devs) > 1:
        raise WtError(f"tensors of one call live on different devices: {sorted(map(str, devs))}")


def _need_cuda(t: torch.Tensor, name: str):
    if not (isinstance(t, torch.Tensor) and t.is_cuda):
        raise WtError(f"{name} must be a CUDA(HIP) tensor: the alignment kernels only run on the GPU")


def require_gpu(device, what="the MI355X alignment path"):
    if torch.device(device).type != "cuda":
        raise WtError(f"{what} needs the model and its tensors on the GPU (there is no CPU fallback)")


def _ptr(t):
    return 0 if t is None else t.data_ptr()


def make_descs(n: int) -> np.ndarray:
    return np.zeros(n, dtype=SEG_DTYPE)


def descs_to_device(descs: np.ndarray, device) -> torch.Tensor:
    raw = torch.from_numpy(descs.view(np.uint8).reshape(-1))
    return raw.to(device, non_blocking=False)


def small_unit(T: int, F: int) -> bool:
    """Mirror of wt_small_unit (csrc/wt_small.h): does a unit of T tokens x F frames take the fused tail kernel of
    wt_align_batch_v3 (column norm, cost[0,0], DTW and backtrack in one workgroup, the matrix in LDS)?  Informational
    (tests, diagnostics): the library decides by itself."""
    if not (1 <= T <= WT_MAX_TOKENS and 1 <= F <= WT_MAX_FRAMES):
        return False
    nw = (T + 63) // 64
    pitch = (F + T + 3) & ~3
    if pitch & 4 == 0:
        pitch += 4
    planes = ((F + 63 + 31) // 32 + 1) * 64 * nw * 8
    bnd = (nw - 1) * (((F + 64 + 32 + 1) & ~1) + 96) * 8 + 16
    return planes + bnd + (T * pitch + 192) * 4 + 64 <= 160 * 1024


def launch_order(shapes):
    """Order in which to hand units to the library: grouped by the cost kernel's F class (ceil(F/256)), longest
    first inside a class.  wt_cost_batch then launches every class over its own units only (wt_cost.hip)."""
    return sorted(range(len(shapes)), key=lambda i: ((shapes[i][1] + 255) // 256, -shapes[i][0], -shapes[i][1], i))


def layout_outputs(descs: np.ndarray):
    """Fill cost/jumps/path offsets (cost slots 16-byte aligned).  Returns totals; the cost total includes the
    16 bytes of read slack wt_dtw_batch asks for (include/wtalign.h)."""
    c = j = p = 0
    for d in descs:
        d["cost_offset"], d["jumps_offset"], d["path_offset"] = c, j, p
        c += (int(d["T"]) * int(d["F"]) + 3) & ~3
        j += int(d["T"]) + 1
        p += int(d["T"]) + int(d["F"]) - 1
    return c + 4, j, p


def cost_batch(qk: torch.Tensor, descs: np.ndarray, descs_dev: torch.Tensor, head_idx: torch.Tensor, cost: torch.Tensor,
               medfilt_width: int = 9, qk_scale: float = 1.0):
    _need_cuda(qk, "qk")
    same_device(qk, descs_dev, head_idx, cost)
    dt = {torch.float32: WT_DTYPE_F32, torch.float16: WT_DTYPE_F16}[qk.dtype]
    with on_device(qk) as st:
        rc = load().wt_cost_batch(qk.data_ptr(), dt, descs.ctypes.data, descs_dev.data_ptr(), len(descs), head_idx.data_ptr(),
                                  head_idx.numel(), medfilt_width, qk_scale, cost.data_ptr(), st)
    _check(rc, "wt_cost_batch")


def dtw_batch(cost: torch.Tensor, descs: np.ndarray, descs_dev: torch.Tensor, jumps: torch.Tensor, path_i=None, path_j=None,
              path_len=None, dist=None, step_pattern: int = WT_STEP_SYMMETRIC1):
    _need_cuda(cost, "cost")
    same_device(cost, descs_dev, jumps, path_i, path_j, path_len, dist)
    with on_device(cost) as st:
        rc = load().wt_dtw_batch_pattern(cost.data_ptr(), descs.ctypes.data, descs_dev.data_ptr(), len(descs), int(step_pattern),
                                         jumps.data_ptr(), _ptr(path_i), _ptr(path_j), _ptr(path_len), _ptr(dist), st)
    _check(rc, "wt_dtw_batch")


def align_batch(qk, descs, descs_dev, head_idx, cost, jumps, path_i=None, path_j=None, path_len=None, dist=None,
                medfilt_width: int = 9, qk_scale: float = 1.0, flags: int = WT_ALIGN_KEEP_COST):
    _need_cuda(qk, "qk")
    same_device(qk, descs_dev, head_idx, cost, jumps, path_i, path_j, path_len, dist)
    dt = {torch.float32: WT_DTYPE_F32, torch.float16: WT_DTYPE_F16}[qk.dtype]
    with on_device(qk) as st:
        rc = load().wt_align_batch_v3(qk.data_ptr(), dt, descs.ctypes.data, descs_dev.data_ptr(), len(descs), head_idx.data_ptr(),
                                      head_idx.numel(), medfilt_width, qk_scale, cost.data_ptr(), jumps.data_ptr(), _ptr(path_i),
                                      _ptr(path_j), _ptr(path_len), _ptr(dist), int(flags), st)
    _check(rc, "wt_align_batch_v3")


def find_start_padding(mel: torch.Tensor) -> torch.Tensor:
    """mel: (B, n_mels, n_cols) fp32 on the GPU -> int32[B] (-1 = None)."""
    _need_cuda(mel, "mel")
    mel = mel.contiguous()
    B, M, C = mel.shape
    out = torch.empty(B, dtype=torch.int32, device=mel.device)
    with on_device(mel) as st:
        rc = load().wt_find_start_padding_batch(mel.data_ptr(), B, M, C, out.data_ptr(), st)
    _check(rc, "wt_find_start_padding_batch")
    return out


def logprob_gather(logits: torch.Tensor, tokens: torch.Tensor, suppress: torch.Tensor | None = None) -> torch.Tensor:
    """logits: (n, V) fp32/fp16 (row stride arbitrary, unit column stride); tokens: int32[n];
    suppress: optional uint8/bool (V,) or (n, V).  Returns fp32[n]."""
    _need_cuda(logits, "logits")
    assert logits.dim() == 2 and logits.stride(1) == 1
    n, V = logits.shape
    dt = {torch.float32: WT_DTYPE_F32, torch.float16: WT_DTYPE_F16}[logits.dtype]
    tokens = tokens.to(device=logits.device, dtype=torch.int32).contiguous()
    out = torch.empty(n, dtype=torch.float32, device=logits.device)
    srows, sp = 0, 0
    if suppress is not None:
        suppress = suppress.to(device=logits.device).to(torch.uint8).contiguous()
        srows = 1 if suppress.dim() == 1 else suppress.shape[0]
        sp = suppress.data_ptr()
    with on_device(logits) as st:
        rc = load().wt_logprob_gather_batch(logits.data_ptr(), dt, logits.stride(0) if n > 1 else V, n, V, tokens.data_ptr(), sp,
                                            srows, out.data_ptr(), st)
    _check(rc, "wt_logprob_gather_batch")
    return out


def logprob_gather_rows(logits: torch.Tensor, row_index: torch.Tensor, tokens: torch.Tensor, out: torch.Tensor | None = None):
    """out[r] = log_softmax(logits[row_index[r]])[tokens[r]].  logits: (n_rows, V) fp32/fp16 with unit column stride;
    row_index, tokens: int32[n_out] on the GPU (rows may repeat).  No (n, V) log-prob matrix is materialised."""
    _need_cuda(logits, "logits")
    assert logits.dim() == 2 and logits.stride(1) == 1
    same_device(logits, row_index, tokens, out)
    assert row_index.dtype == torch.int32 and tokens.dtype == torch.int32 and row_index.numel() == tokens.numel()
    n = row_index.numel()
    if out is None:
        out = torch.empty(n, dtype=torch.float32, device=logits.device)
    assert out.dtype == torch.float32 and out.numel() >= n and out.is_contiguous()
    dt = {torch.float32: WT_DTYPE_F32, torch.float16: WT_DTYPE_F16}[logits.dtype]
    with on_device(logits) as st:
        rc = load().wt_logprob_gather_rows(logits.data_ptr(), dt, logits.stride(0), row_index.data_ptr(), n, logits.shape[1],
                                           tokens.data_ptr(), out.data_ptr(), st)
    _check(rc, "wt_logprob_gather_rows")
    return out


DIGEST_WORDS = 8          # floats per record of wt_logprob_digest_streams (include/wtalign.h)


def logprob_digest_streams(rows: torch.Tensor, tokens: torch.Tensor, ring_index: torch.Tensor, digest: torch.Tensor,
                           slice_ring: torch.Tensor | None, ring_row: int, aux_tokens, slice_begin: int):
    """One decoder call of g streams: rows (g, V) fp32 (any row stride), tokens (g,) int32 / int64 (any stride): the token
    sampled from each row.  digest: (n_blocks, ring_rows, 8) fp32, slice_ring: (n_blocks, ring_rows, V - slice_begin) fp32
    or None; batch row r writes block ring_index[r], row ring_row.  See wt_logprob_digest_streams."""
    _need_cuda(rows, "rows")
    assert rows.dim() == 2 and rows.stride(1) == 1 and rows.dtype == torch.float32
    g, V = rows.shape
    assert tokens.dim() == 1 and tokens.numel() == g and tokens.dtype in (torch.int32, torch.int64)
    assert ring_index.dtype == torch.int32 and ring_index.numel() == g
    assert digest.is_contiguous() and digest.dtype == torch.float32 and digest.shape[-1] == DIGEST_WORDS
    same_device(rows, tokens, ring_index, digest, slice_ring)
    if slice_ring is not None:
        assert slice_ring.is_contiguous() and slice_ring.dtype == torch.float32 and \
            slice_ring.shape == (digest.shape[0], digest.shape[1], V - slice_begin), (slice_ring.shape, digest.shape, V, slice_begin)
    aux = (ctypes.c_int32 * max(1, len(aux_tokens)))(*[int(t) for t in aux_tokens])
    with on_device(rows) as st:
        rc = load().wt_logprob_digest_streams(rows.data_ptr(), rows.stride(0) if g > 1 else V, g, V, tokens.data_ptr(),
                                              1 if tokens.dtype == torch.int64 else 0, tokens.stride(0) if g > 1 else 1,
                                              ring_index.data_ptr(), digest.shape[1], int(ring_row), aux, len(aux_tokens),
                                              int(slice_begin), digest.data_ptr(),
                                              slice_ring.data_ptr() if slice_ring is not None else None, st)
    _check(rc, "wt_logprob_digest_streams")


def qk_rows_batch(q_layers, k_layers, sel_layer, sel_head, sel_slot, ring: torch.Tensor, row_begin=None, row_end=None,
                  ring_row0: int = 0):
    """ring[b, slot, ring_row0 + r, :] = the QK logits row of query r for every selected (layer, head), every window b.
    q_layers / k_layers: per hooked layer (B, n_q, D) / (B, n_ctx, D) projections; ring: (B, n_slots, rows, n_ctx)."""
    import ctypes as C
    q0, k0 = q_layers[0], k_layers[0]
    _need_cuda(q0, "q")
    same_device(*q_layers, *k_layers, sel_layer, sel_head, sel_slot, ring, row_begin, row_end)
    B, n_q, D = q0.shape
    n_ctx = k0.shape[1]
    for q, k in zip(q_layers, k_layers):
        assert q.shape == q0.shape and k.shape == k0.shape and q.dtype == q0.dtype == k.dtype, (q.shape, k.shape)
        assert q.stride(2) == 1 and q.stride(1) == D and k.stride(2) == 1 and k.stride(1) == D
        assert q.stride(0) == q0.stride(0) and k.stride(0) == k0.stride(0)
    assert ring.dim() == 4 and ring.shape[0] == B and ring.shape[3] == n_ctx and ring[0].is_contiguous()
    n_heads = D // 64
    qp = (C.c_void_p * len(q_layers))(*[q.data_ptr() for q in q_layers])
    kp = (C.c_void_p * len(k_layers))(*[k.data_ptr() for k in k_layers])
    dt = {torch.float32: WT_DTYPE_F32, torch.float16: WT_DTYPE_F16}
    with on_device(q0) as st:
        rc = load().wt_qk_rows_batch(qp, kp, len(q_layers), dt[q0.dtype], B, n_q, q0.stride(0), k0.stride(0), n_ctx, D,
                                     D // n_heads, float(D // n_heads) ** -0.25, sel_layer.data_ptr(), sel_head.data_ptr(),
                                     sel_slot.data_ptr(), sel_layer.numel(), _ptr(row_begin), _ptr(row_end), ring.data_ptr(),
                                     dt[ring.dtype], ring.stride(0), ring.shape[2], int(ring_row0), st)
    _check(rc, "wt_qk_rows_batch")


def logmel(pcm: torch.Tensor, mel_fb: torch.Tensor, n_valid_samples: torch.Tensor | None = None, n_frames: int = 3000,
           with_padding: bool = False, launch=None):
    """pcm: (B, n_samples) fp32; mel_fb: (n_mels, 201) fp32.  Returns (mel (B,n_mels,n_frames), gmax (B,)) -- and, with
    ``with_padding``, find_start_padding of every window (int32[B], -1 = None) by a one-wave-per-window pass that starts at the last valid column.
    ``launch``: a callable ``launch(fn)`` that runs ``fn(stream_handle)`` on a stream of its choice (pipeline.StageSet.run:
    the outputs are allocated here, on the caller's current stream; the kernels go where the schedule puts them)."""
    _need_cuda(pcm, "pcm")
    pcm = pcm.contiguous().float()
    mel_fb = mel_fb.to(pcm.device).contiguous().float()
    B, N = pcm.shape
    M = mel_fb.shape[0]
    assert mel_fb.shape[1] == 201
    mel = torch.empty((B, M, n_frames), dtype=torch.float32, device=pcm.device)
    gmax = torch.empty(B, dtype=torch.float32, device=pcm.device)
    nv = None if n_valid_samples is None else n_valid_samples.to(device=pcm.device, dtype=torch.int32).contiguous()
    if with_padding:
        pad = torch.empty(B, dtype=torch.int32, device=pcm.device)

        def go(st):
            _check(load().wt_logmel_pad_batch(pcm.data_ptr(), B, N, _ptr(nv), mel_fb.data_ptr(), M, n_frames, mel.data_ptr(),
                                              gmax.data_ptr(), pad.data_ptr(), st), "wt_logmel_pad_batch")
        with on_device(pcm) as st:
            if launch is not None:
                launch(go)
            else:
                go(st)
        return mel, gmax, pad
    assert launch is None, "launch=: only with with_padding=True"
    with on_device(pcm) as st:
        rc = load().wt_logmel_batch(pcm.data_ptr(), B, N, _ptr(nv), mel_fb.data_ptr(), M, n_frames, mel.data_ptr(),
                                    gmax.data_ptr(), st)
    _check(rc, "wt_logmel_batch")
    return mel, gmax
