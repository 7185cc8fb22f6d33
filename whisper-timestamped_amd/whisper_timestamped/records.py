"""transcribe() dictionaries on their way from the ranks to rank 0 (sharding.py): one byte record per recording in ONE
fixed-size tensor gather, decoded on rank 0 only when somebody reads it.

A result dictionary (whisper's + the reference's additions: /root/reference/whisper_timestamped/transcribe.py:959-997,
schema tests/golden/json_schema.json) is ~10 KB of small Python objects -- ~140 word dictionaries per 30 s of speech.
``dist.gather_object`` (rounds 1-5) pickles a rank's whole list and REBUILDS every object of every rank on rank 0 before
it returns: 0.26 ms per recording measured on rank 0 (tools/measure_result_gather.py; with 8 ranks that is 2 ms of rank 0's
serial time per recording it decodes itself, against ~16 ms of compute).  Rebuilding the objects is the cost, not the
bytes: a hand-written flat layout (ints / floats / text arrays, tried first this round) packs and unpacks in Python and
lost to the C pickler by 2-4x.  So the record IS the pickle of one recording's dictionary, and what changes is WHEN it is
decoded:

    every rank   record_i = pickle.dumps(result_i)            (C speed, in parallel on the ranks)
    one gather   [n, (index, n_bytes) x n, bytes ...] as int32, padded to the longest rank's message (fixed size, like
                 sharding.ResultGatherer's step records); the sizes travel first in one small all-gather
    rank 0       PackedResults: the bytes + an index.  ``table.dict(i)`` decodes ONE recording; a writer that streams JSON
                 files out holds one dictionary at a time; ``table.dicts()`` decodes all (what results="dicts" returns).
"""
from __future__ import annotations

import pickle

import numpy as np
import torch


class PackedResults:
    """Byte records of many recordings' results + where each one lies; dictionaries are built on demand."""

    def __init__(self, indices, blob, spans):
        self.indices = list(indices)            # recording index of every record, in storage order
        self._pos = {r: k for k, r in enumerate(self.indices)}
        self.blob = blob                        # uint8 array
        self.spans = spans                      # (n, 2) int64: [begin, end) of every record in `blob`

    def __len__(self):
        return len(self.indices)

    def nbytes(self, recording: int) -> int:
        b, e = self.spans[self._pos[recording]]
        return int(e - b)

    def dict(self, recording: int) -> dict:
        b, e = (int(x) for x in self.spans[self._pos[recording]])
        return pickle.loads(self.blob[b:e].tobytes())

    def dicts(self):
        return [self.dict(r) for r in sorted(self.indices)]


def pack_many(pairs) -> np.ndarray:
    """[(recording index, result dict)] -> one int32 message: [n, (index, n_bytes) x n, the records' bytes padded to 4]."""
    recs = [(int(i), pickle.dumps(r, protocol=pickle.HIGHEST_PROTOCOL)) for i, r in pairs]
    header = [len(recs)]
    for i, b in recs:
        header.extend((i, len(b)))
    body = b"".join(b for _, b in recs)
    body += b"\0" * (-len(body) % 4)
    return np.concatenate([np.asarray(header, dtype=np.int32), np.frombuffer(body, dtype=np.int32)])


def split_message(msg: np.ndarray):
    """One rank's pack_many message (padding behind it allowed) -> (indices, blob uint8, spans)."""
    n = int(msg[0])
    head = msg[1:1 + 2 * n].reshape(n, 2).astype(np.int64)
    end = np.cumsum(head[:, 1])
    blob = msg[1 + 2 * n:].view(np.uint8)[:int(end[-1]) if n else 0]
    return head[:, 0].tolist(), blob, np.stack([end - head[:, 1], end], axis=1)


def table_of(messages) -> PackedResults:
    """The messages of several ranks (each: pack_many's array, possibly padded) as one table."""
    indices, blobs, spans, off = [], [], [], 0
    for msg in messages:
        idx, blob, sp = split_message(np.ascontiguousarray(msg))
        indices += idx
        blobs.append(blob)
        spans.append(sp + off)
        off += blob.size
    return PackedResults(indices, np.concatenate(blobs) if blobs else np.zeros(0, np.uint8),
                         np.concatenate(spans) if spans else np.zeros((0, 2), dtype=np.int64))


def gather_packed(dist, pairs, device) -> PackedResults | None:
    """Every rank's [(recording index, dict)] to rank 0 as ONE fixed-size int32 tensor gather.  Rank 0 gets a
    PackedResults, the others None."""
    rank, world = dist.get_rank(), dist.get_world_size()
    msg = torch.from_numpy(pack_many(pairs))
    sizes = [torch.zeros(1, dtype=torch.int64, device=device) for _ in range(world)]
    dist.all_gather(sizes, torch.tensor([msg.numel()], dtype=torch.int64, device=device))
    sizes = [int(s.item()) for s in sizes]
    longest = max(sizes)
    send = torch.zeros(longest, dtype=torch.int32, device=device)
    send[:msg.numel()].copy_(msg)
    recv = [torch.empty(longest, dtype=torch.int32, device=device) for _ in range(world)] if rank == 0 else None
    dist.gather(send, recv, dst=0)
    if rank != 0:
        return None
    return table_of([recv[r][:sizes[r]].cpu().numpy() for r in range(world)])
