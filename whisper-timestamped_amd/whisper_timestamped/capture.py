"""Device-side data plane of the decode-time hooks.

The reference's hooks move data to the host once per token:
``hook_attention_weights`` does ``w[:, :, -1:, :].cpu()`` for every hooked layer
(/root/reference/whisper_timestamped/transcribe.py:783-793) and
``hook_output_logits`` keeps one full ``log_softmax`` vector (V floats) per step
(:849-881).  Here both live in preallocated device rings:

* ``QKCaptureRing`` -- (A_sel, capacity, n_ctx): only the alignment heads are
  stored; the hook calls ``wt_capture_rows`` (one async launch, no sync).
  A segment is a list of ring rows; contiguous rows are handed to the cost
  kernel as a strided view (no copy).
* ``LogitsRing`` -- (capacity, V) fp32 *filtered logits* per step.  The
  log-softmax is never materialised: the chosen-token log-probabilities of a
  whole window are produced by one ``wt_logprob_gather_batch`` at flush time.
"""
from __future__ import annotations

import torch

from . import _lib


def layer_head_slots(alignment_heads_pairs, n_hooked_layers: int, n_heads: int):
    """Per hooked layer: (heads, slots).  Slot order = order of the reference's
    head stacking: rows of ``alignment_heads.indices().T`` (COO, layer-major),
    or layer-major all heads when no alignment heads are known
    (transcribe.py:1542-1545)."""
    per_layer = [([], []) for _ in range(n_hooked_layers)]
    if alignment_heads_pairs is None:
        slot = 0
        for l in range(n_hooked_layers):
            for h in range(n_heads):
                per_layer[l][0].append(h)
                per_layer[l][1].append(slot)
                slot += 1
        return per_layer, slot
    for slot, (l, h) in enumerate(alignment_heads_pairs):
        assert 0 <= l < n_hooked_layers and 0 <= h < n_heads, f"alignment head ({l},{h}) outside the hooked layers"
        per_layer[l][0].append(h)
        per_layer[l][1].append(slot)
    return per_layer, len(alignment_heads_pairs)


class QKCaptureRing:
    def __init__(self, device, alignment_heads_pairs, n_hooked_layers: int, n_heads: int, n_ctx: int = 1500,
                 capacity: int = 448, dtype=torch.float32):
        per_layer, self.n_slots = layer_head_slots(alignment_heads_pairs, n_hooked_layers, n_heads)
        self.device = torch.device(device)
        self.n_ctx, self.capacity, self.n_heads = n_ctx, capacity, n_heads
        self.buf = torch.zeros((self.n_slots, capacity, n_ctx), dtype=dtype, device=device)
        self._dt = {torch.float32: _lib.WT_DTYPE_F32, torch.float16: _lib.WT_DTYPE_F16}[dtype]
        self._heads = [torch.tensor(h, dtype=torch.int32, device=device) for h, _ in per_layer]
        self._slots = [torch.tensor(s, dtype=torch.int32, device=device) for _, s in per_layer]
        self._lib = _lib.load()

    def write(self, layer_index: int, qk: torch.Tensor, row: int):
        """qk: (1, H, n_q, n_ctx) QK logits of one layer; stores its last query row at ring row `row`."""
        _lib._need_cuda(qk, "qk")
        assert qk.dim() == 4 and qk.shape[0] == 1 and qk.shape[1] == self.n_heads and qk.shape[3] == self.n_ctx, qk.shape
        if not qk.is_contiguous():
            qk = qk.contiguous()
        heads = self._heads[layer_index]
        if heads.numel() == 0:
            return
        dt = {torch.float32: _lib.WT_DTYPE_F32, torch.float16: _lib.WT_DTYPE_F16}[qk.dtype]
        _lib.same_device(qk, self.buf)
        with _lib.on_device(self.buf) as st:
            rc = self._lib.wt_capture_rows(qk.data_ptr(), dt, qk.shape[1], qk.shape[2], self.n_ctx, heads.data_ptr(),
                                           self._slots[layer_index].data_ptr(), heads.numel(), self.buf.data_ptr(), self._dt,
                                           self.capacity, int(row), st)
        _lib._check(rc, "wt_capture_rows")

    def write_from_projections(self, layer_index: int, q: torch.Tensor, k: torch.Tensor, row0: int, n_rows: int = 1):
        """The same rows from the cross-attention projections (wt_qk_rows): q = cross_attn.query(x) (1, n_q, D), of
        which the LAST n_rows query rows are used; k = cross_attn.key(xa) (1, n_ctx, D).  The backend can then stay
        on its fused attention path (no whisper.model.disable_sdpa())."""
        _lib._need_cuda(q, "q")
        heads = self._heads[layer_index]
        if heads.numel() == 0:
            return
        assert q.dim() == 3 and k.dim() == 3 and q.shape[0] == 1 and k.shape[0] == 1 and k.shape[1] == self.n_ctx, (q.shape, k.shape)
        assert q.dtype == k.dtype and q.shape[2] == k.shape[2]
        d_model = q.shape[2]
        head_dim = d_model // self.n_heads
        qr = q[0, q.shape[1] - n_rows:]
        kr = k[0]
        if not qr.is_contiguous():
            qr = qr.contiguous()
        if not kr.is_contiguous():
            kr = kr.contiguous()
        dt = {torch.float32: _lib.WT_DTYPE_F32, torch.float16: _lib.WT_DTYPE_F16}[q.dtype]
        _lib.same_device(q, k, self.buf)
        with _lib.on_device(self.buf) as st:
            rc = self._lib.wt_qk_rows(qr.data_ptr(), kr.data_ptr(), dt, n_rows, self.n_ctx, d_model, head_dim,
                                      float(head_dim) ** -0.25, heads.data_ptr(), self._slots[layer_index].data_ptr(),
                                      heads.numel(), self.buf.data_ptr(), self._dt, self.capacity, int(row0), st)
        _lib._check(rc, "wt_qk_rows")

    def used_layers(self):
        """Hooked-layer indices that own at least one selected head (whisper-small: 4 of 12, large-v3: 10 of 32): the only
        layers whose projections have to be observed."""
        return [l for l, h in enumerate(self._heads) if h.numel() > 0]

    def write_all_layers(self, q_layers, k_layers, row: int):
        """ONE launch for every layer that owns a selected head (wt_qk_rows_batch with a single window): the LAST query
        row of each layer's q (1, n_q, D) against its K (1, n_ctx, D) -> ring row `row` of every selected head.  This
        runs once per decoded token: everything that does not change from token to token (head tables, argument
        arrays) is built once, a call only refreshes the layer pointers."""
        if self.n_slots == 0:
            return
        import ctypes as C
        fast = getattr(self, "_fast", None)
        if fast is None:
            used = self.used_layers()
            sel = [(i, h, s) for i, l in enumerate(used) for h, s in zip(self._heads[l].tolist(), self._slots[l].tolist())]
            fast = self._fast = dict(
                used=used, qp=(C.c_void_p * len(used))(), kp=(C.c_void_p * len(used))(),
                sel=tuple(torch.tensor([x[i] for x in sel], dtype=torch.int32, device=self.device) for i in range(3)), n_sel=len(sel))
        q0, k0 = q_layers[fast["used"][0]], k_layers[fast["used"][0]]
        D = q0.shape[2]
        assert q0.dim() == 3 and q0.shape[0] == 1 and q0.stride(2) == 1 and q0.stride(1) == D and k0.shape[1] == self.n_ctx \
            and k0.stride(1) == D and k0.dtype == q0.dtype and D == self.n_heads * 64, (q0.shape, k0.shape, q0.dtype)
        last = (q0.shape[1] - 1) * D * q0.element_size()           # byte offset of the last query row
        for i, l in enumerate(fast["used"]):
            fast["qp"][i] = q_layers[l].data_ptr() + last
            fast["kp"][i] = k_layers[l].data_ptr()
        dt = _lib.WT_DTYPE_F32 if q0.dtype == torch.float32 else _lib.WT_DTYPE_F16
        sl, sh, ss = fast["sel"]
        with _lib.on_device(self.buf) as st:
            rc = self._lib.wt_qk_rows_batch(fast["qp"], fast["kp"], len(fast["used"]), dt, 1, 1, D, self.n_ctx * D, self.n_ctx, D,
                                            64, 64.0 ** -0.25, sl.data_ptr(), sh.data_ptr(), ss.data_ptr(), fast["n_sel"], 0, 0,
                                            self.buf.data_ptr(), self._dt, self.buf.numel(), self.capacity, int(row), st)
        _lib._check(rc, "wt_qk_rows_batch")

    def rows(self, rows) -> torch.Tensor:
        """(A_sel, len(rows), n_ctx): a strided VIEW when the rows are consecutive, else a device gather."""
        rows = list(rows)
        if rows and rows == list(range(rows[0], rows[0] + len(rows))):
            return self.buf[:, rows[0]:rows[0] + len(rows)]
        idx = torch.tensor(rows, dtype=torch.long, device=self.device)
        return self.buf.index_select(1, idx)


class LogitsRing:
    def __init__(self, device, n_vocab: int, capacity: int = 448):
        self.buf = torch.empty((capacity, n_vocab), dtype=torch.float32, device=device)
        self.n = 0

    def reset(self):
        self.n = 0

    def append(self, logits_row: torch.Tensor):
        """logits_row: (V,) or (1,V) filtered logits (suppressed entries = -inf) of the step just computed."""
        self.buf[self.n].copy_(logits_row.reshape(-1))
        self.n += 1

    def next_row(self) -> torch.Tensor:
        """(1, V) view of the row the next step goes to: the caller computes the step's logits straight INTO the ring
        (``torch.matmul(..., out=row)``), filters them in place, then calls ``commit()`` -- no per-token copy."""
        return self.buf[self.n:self.n + 1]

    def commit(self):
        self.n += 1

    def __len__(self):
        return self.n

    def argmax(self, row: int, lo: int = 0) -> int:
        """argmax of log_softmax(row)[lo:] (+lo).  log_softmax is monotone: taken on the logits."""
        if row < 0:
            row += self.n
        return int(torch.argmax(self.buf[row, lo:]).item()) + lo

    def gather(self, tokens) -> torch.Tensor:
        """fp32[n]: log_softmax(row k)[tokens[k]] for the first len(tokens) rows (one kernel, no (n,V) matrix)."""
        n = len(tokens)
        assert n <= self.n
        tok = torch.as_tensor(tokens, dtype=torch.int32)
        return _lib.logprob_gather(self.buf[:n], tok)
