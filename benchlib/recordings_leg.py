"""N > 1 only: ragged recordings dealt to the ranks (sharding.transcribe_recordings), decoder streams within a rank."""
import os
import sys
import time

import numpy as np
import torch

from .common import log, make_emitter
from wordgaps import NO_GAPS, PARITY_FAILURES, gaps_ok_between_batch_sizes, gaps_report, merge_gaps, parity_flag, word_gaps, words_of


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
        wins.append([H.ragged_window(rs, int(sec * 50), TS0, EOT, *((40, 160) if not dry else (8, 16)))])
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
        # results="packed": one byte record per recording in one fixed-size tensor gather; rank 0 decodes what it reads
        res = transcribe_recordings(model, audios, dist=dist, broadcast_weights=dry, streams=n_streams, on_batch=on_batch,
                                    results="packed", **opts)
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
        assert len(res) == len(audios) and all(res.nbytes(i) > 0 for i in range(len(audios)))
        t_dec = time.perf_counter()
        every_result = res.dicts()                         # (what a caller that wants every dictionary at once would pay on rank 0)
        t_dec = time.perf_counter() - t_dec
        assert all(len(r["segments"]) > 0 for r in every_result)
        worst = NO_GAPS
        picks = sorted({0, len(audios) // 2, len(audios) - 1})
        for k in picks:                                    # recordings other ranks decoded, against one stream here
            set_script(Script(wins[k]))
            try:
                alone = wt.transcribe(model, audios[k], **opts)
            finally:
                set_script(None)
            worst = merge_gaps(worst, word_gaps(words_of(res.dict(k)), words_of(alone), f"recording {k} (another rank) vs one stream"))
        parity_flag(gaps_ok_between_batch_sizes(worst), "transcribe_recordings", gaps_report(worst))
        emit({"parity_failures": list(PARITY_FAILURES), "what": "sharding.transcribe_recordings: ragged recordings (U[5, 30] s, own transcripts) dealt to the ranks, "
                      f"{n_streams} decoder streams per rank, weights broadcast from rank 0 (every other rank started from "
                      "perturbed weights), one byte record per recording gathered to rank 0 in one fixed-size tensor gather",
              "ranks": world, "recordings": len(audios), "recordings_per_rank": per_rank, "audio_seconds": round(sum(secs), 1),
              "seconds": round(el, 3), "audio_s_per_s": round(sum(secs) / el, 1), "scaling": "weak",
              "per_rank_seconds": every, "results": "packed (records.py): decoded on demand",
              "decoding_every_result_on_rank_0_seconds": round(t_dec, 4), "backend": "gloo (dry run)" if dry else "rccl",
              "parity_vs_1_stream_on_rank_0": gaps_report(worst, {"recordings_compared": picks})})
    dist.barrier()
    dist.destroy_process_group()

