"""TEST-ONLY stand-in for the HIP kernels, built from the CPU oracle.

The product has no CPU path (whisper_timestamped._lib refuses CPU tensors).  To
test the HOST logic (hook state machine, strategy drivers, confidence glue,
post-processors) in the GPU-less build container, ``install(monkeypatch)``
replaces the five kernel entry points the host layer calls by oracle-backed
CPU functions.  Nothing outside tests/ may import this module.
"""
import numpy as np
import torch

from oracle import align_ref as O


def install(monkeypatch):
    from whisper_timestamped import _lib, alignment, batched, capture, efficient
    monkeypatch.setattr(efficient, "GPU_FRONT_END", False)      # the backend's own torch.stft on the CPU
    monkeypatch.setattr(batched, "SCHEDULE", "serial")          # (the default) no HIP streams on the CPU: program order
    monkeypatch.setattr(efficient, "FUSED_ATTENTION", False)    # qk observed on the unfused path, as in the reference

    monkeypatch.setattr(_lib, "require_gpu", lambda device, what="": None)
    monkeypatch.setattr(_lib, "_need_cuda", lambda t, name: None)

    def write(self, layer_index, qk, row):
        heads, slots = self._heads[layer_index].long(), self._slots[layer_index].long()
        if heads.numel():
            self.buf[slots, row] = qk[0, heads, -1].to(self.buf.dtype)
    monkeypatch.setattr(capture.QKCaptureRing, "write", write)

    def logprob_gather(logits, tokens, suppress=None):
        return O.token_logprob_gather_ref(logits.float(), np.asarray(tokens, dtype=np.int64),
                                          None if suppress is None else suppress.bool())
    monkeypatch.setattr(_lib, "logprob_gather", logprob_gather)

    def find_start_padding(mel):
        out = []
        for b in range(mel.shape[0]):
            r = O.find_start_padding_ref(mel[b:b + 1])
            out.append(-1 if r is None else int(r))
        return torch.tensor(out, dtype=torch.int32)
    monkeypatch.setattr(_lib, "find_start_padding", find_start_padding)

    def logmel(pcm, mel_fb, n_valid_samples=None, n_frames=3000, with_padding=False, launch=None):
        B = pcm.shape[0]
        M = mel_fb.shape[0]
        mel = torch.zeros((B, M, n_frames))
        gmax = torch.zeros(B)
        for b in range(B):
            n = pcm.shape[1] if n_valid_samples is None else int(n_valid_samples[b])
            m = O.log_mel_spectrogram_ref(pcm[b, :n], M)
            k = min(m.shape[-1], n_frames)
            mel[b, :, :k] = m[:, :k]
        if with_padding:
            return mel, gmax, find_start_padding(mel)
        return mel, gmax
    monkeypatch.setattr(_lib, "logmel", logmel)

    def launch(self):
        # like the real launch, this reads the units' QK rows NOW (later tokens may overwrite the capture ring)
        self._launched = True
        self.extra = torch.zeros(self.extra_words, dtype=torch.int32) if self.extra_words else None
        self._numbers = []
        for u in self.units:
            sel = u.qk[:, :, u.start_token:u.end_token].contiguous().float().numpy()
            pad = u.pad_from if u.pad_from >= 0 else None
            cost = O.cost_matrix_ref(sel, self.medfilt_width, self.qk_scale, pad, u.start_token)
            r = O.dtw_ref(cost, step_pattern=self.step_pattern)
            jumps = O.jumps_from_path(r.index1s, r.index2s).astype(np.int64)
            self._numbers.append((jumps, O.jumps_start_ref(cost, jumps) if u.detect_disfluencies else None))
        return self

    def fetch(self):
        if not self._launched:
            self.launch()
        return self

    def collect(self):
        self.fetch()
        out = [alignment.finish_unit(u, jumps, starts) for u, (jumps, starts) in zip(self.units, self._numbers)]
        self.extra_host = self.extra.numpy() if self.extra is not None else None
        return out
    monkeypatch.setattr(alignment.AlignmentBatch, "launch", launch)
    monkeypatch.setattr(alignment.AlignmentBatch, "fetch", fetch)
    monkeypatch.setattr(alignment.AlignmentBatch, "collect", collect)

    # batched naive strategy (whisper_timestamped/batched.py)
    def qk_rows_batch(q_layers, k_layers, sel_layer, sel_head, sel_slot, ring, row_begin=None, row_end=None, ring_row0=0):
        B, n_q, D = q_layers[0].shape
        hd = 64
        scale = hd ** -0.25
        for l, h, s in zip(sel_layer.tolist(), sel_head.tolist(), sel_slot.tolist()):
            q = (q_layers[l][:, :, h * hd:(h + 1) * hd] * scale)
            k = (k_layers[l][:, :, h * hd:(h + 1) * hd] * scale)
            ring[:, s, ring_row0:ring_row0 + n_q] = (q @ k.transpose(1, 2)).float().to(ring.dtype)
    monkeypatch.setattr(_lib, "qk_rows_batch", qk_rows_batch)

    def logprob_gather_rows(logits, row_index, tokens, out=None):
        lp = O.token_logprob_gather_ref(logits[row_index.long()].float(), tokens.numpy().astype(np.int64))
        if out is None:
            return lp
        out[:lp.numel()].copy_(lp)
        return out
    monkeypatch.setattr(_lib, "logprob_gather_rows", logprob_gather_rows)
