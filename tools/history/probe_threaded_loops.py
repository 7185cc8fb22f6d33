#!/usr/bin/env python3
"""Can several SMALL decoder loops run side by side in ONE process?  (Ragged long form: 35 of 42 decoder loops carry one
stream, each ~110 steps x ~3 ms of host time, the GPU nearly idle.)  The backend's KV-cache hooks sit on the model's
modules, so two loops cannot share a module tree -- but a replica of the module tree that SHARES the parameter tensors
costs no memory and has its own hooks.  K threads, each with its own replica and HIP stream, each running n_loops
one-stream decoder loops (the backend's own DecodingTask._main_loop, greedy, scripted to 110 steps); torch releases the
GIL inside every ATen call, so the threads' launches can overlap.      python tools/probe_threaded_loops.py"""
import copy
import json
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "whisper-timestamped_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch  # noqa: E402
import many_helper as H  # noqa: E402
import whisper_double as W  # noqa: E402

W.install()


def replica(model):
    """A copy of the module tree whose parameters and buffers ARE the original's tensors (no memory), with its own hooks."""
    memo = {}
    for t in list(model.parameters()) + list(model.buffers()):
        memo[id(t)] = t
    return copy.deepcopy(model, memo)


def main():
    dev = torch.device("cuda", 0)
    model = H.load_base(dev)
    n_steps = 110
    g = torch.Generator().manual_seed(3)
    mel = (torch.randn((1, 80, 3000), generator=g) * 0.1).to(dev)

    def one_loop(m, sample_len=n_steps):
        task = W.decoding.DecodingTask(m, W.DecodingOptions(language="en", fp16=False, sample_len=sample_len, suppress_blank=False))
        # never sample eot: the loop runs its sample_len steps
        task.logit_filters.append(type("NoEot", (), {"apply": staticmethod(lambda logits, tokens: logits.__setitem__((slice(None), task.tokenizer.eot), -float("inf")))})())
        feats = task._get_audio_features(mel)
        tokens = torch.tensor([list(task.initial_tokens)], device=dev)
        with torch.no_grad():
            out, _, _ = task._main_loop(feats, tokens)
        return out.shape[-1]

    one_loop(model)
    torch.cuda.synchronize()
    results = []
    for K in (1, 2, 3, 4, 6, 8):
        loops_per_thread = 2
        reps = [replica(model) for _ in range(K)]
        streams = [torch.cuda.Stream(device=dev) for _ in range(K)]
        lens = [None] * K

        def work(k):
            with torch.cuda.stream(streams[k]):
                for _ in range(loops_per_thread):
                    lens[k] = one_loop(reps[k])
                streams[k].synchronize()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        ths = [threading.Thread(target=work, args=(k,)) for k in range(K)]
        for t in ths:
            t.start()
        for t in ths:
            t.join()
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        rec = {"threads": K, "loops": K * loops_per_thread, "steps_per_loop": n_steps, "seconds": round(el, 3),
               "ms_per_loop_step_per_thread": round(1e3 * el / (loops_per_thread * n_steps), 3),
               "loops_per_second": round(K * loops_per_thread / el, 2)}
        print(json.dumps(rec), flush=True)
        results.append(rec)
    base = results[0]["loops_per_second"]
    print("speed-up over one thread:", {r["threads"]: round(r["loops_per_second"] / base, 2) for r in results})


if __name__ == "__main__":
    main()
