#!/usr/bin/env python3
"""What does the result gather of sharding.transcribe_recordings cost at N = 8, without an 8-GPU node?  N gloo ranks on
the host, every rank holds R REAL transcribe() dictionaries (the reference-generated outputs of tests/golden/
transcribe_cases.json, cycled; ~60-220 words each), rank 0 gathers them the way the product does (ONE
torch.distributed.gather_object per job) and sorts them back into the caller's order.  Reported: seconds per job and per
recording on rank 0, bytes pickled per recording; DESIGN.md section 8 sets them against one rank's compute seconds for the same
R recordings on the MI355X (bench.py's ragged legs).
    python tools/measure_result_gather.py [ranks=8] [recordings_per_rank=32]"""
import json
import os
import pickle
import sys
import time

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def results_pool():
    cases = json.load(open(os.path.join(ROOT, "tests", "golden", "transcribe_cases.json")))
    pool = [c["expected"] for c in cases if sum(len(s["words"]) for s in c["expected"]["segments"]) >= 40]
    return pool


def worker(rank, world, per_rank, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)                    # (N ranks share this host's cores; a GPU box gives every rank its share)
    pool = results_pool()
    # (distinct objects: pickle would send a repeated dictionary once and a back-reference afterwards)
    mine = [(rank * per_rank + k, json.loads(json.dumps(pool[(rank * per_rank + k) % len(pool)]))) for k in range(per_rank)]
    sys.path.insert(0, os.path.join(ROOT, "whisper-timestamped_amd"))
    from whisper_timestamped.sharding import collect_results
    report = {}
    for mode in ("pickle", "dicts", "packed"):
        times = []
        for rep in range(6):
            dist.barrier()
            t0 = time.perf_counter()
            got = collect_results(dist, mine, world * per_rank, "cpu", mode)
            times.append(time.perf_counter() - t0)
            if rank == 0:
                assert len(got) == world * per_rank
        report[mode] = sorted(times[1:])[len(times[1:]) // 2]
    if rank == 0:
        words = [sum(len(s["words"]) for s in r["segments"]) for _, r in mine]
        from whisper_timestamped import records
        json.dump({"ranks": world, "recordings_per_rank": per_rank, "recordings": world * per_rank,
                   "words_per_recording_mean": round(sum(words) / len(words), 1),
                   "pickled_bytes_per_recording_mean": int(sum(len(pickle.dumps(r)) for _, r in mine) / len(mine)),
                   "packed_bytes_per_recording_mean": int(records.pack_many(mine).nbytes / len(mine)),
                   "rank0_seconds_per_job": {k: round(v, 5) for k, v in report.items()},
                   "rank0_microseconds_per_recording": {k: round(1e6 * v / (world * per_rank), 1) for k, v in report.items()},
                   "what": {"pickle": "dist.gather_object (rounds 1-5)", "dicts": "one byte record per recording, one tensor gather, every "
                            "record decoded on rank 0", "packed": "the same gather, records decoded on demand (none here)"},
                   "backend": "gloo (host memory, this container)", "cpus": os.cpu_count()}, open(out, "w"))
    dist.destroy_process_group()


if __name__ == "__main__":
    world = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    per_rank = int(sys.argv[2]) if len(sys.argv) > 2 else 32
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    out = f"/tmp/wt_gather_{os.getpid()}.json"
    mp.spawn(worker, args=(world, per_rank, port, out), nprocs=world, join=True)
    print(open(out).read())
    os.unlink(out)
