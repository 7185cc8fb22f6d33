#!/usr/bin/env python3
"""bench.py -- whisper-timestamped alignment hot path on MI355X.

One "step" = one pass of the hot path over one batch of synthetic input that
is already resident in HBM:  log-mel STFT front end -> padding detector ->
local-cost construction (head select + median + softmax + head mean + column
norm) -> DTW + backtrack + jumps -> chosen-token log-softmax gather
(confidence), then an async copy of the (KB-sized) jumps/log-probs to the host.

Workload at N=1 (BASELINE.json configs[1]): whisper-base, a batch of 32
synthetic 30 s chunks; every chunk is one full-window alignment unit
(A=8 alignment heads, T=224 tokens, F=1500 frames: the reference's
trust_whisper_timestamps=False shape, SURVEY.md 8(d) "K-full"), V=51865.
For N>1 every rank owns its own 32 chunks (units are independent: weak
scaling, no data-path collective); the per-step result records are gathered
to rank 0 over RCCL, which is where the reference assembles words.

The timed region is the PRODUCT's scheduler in a loop:
``whisper_timestamped.pipeline.HotPathPipeline.submit(batch)`` -- two buffer
sets in flight (--pipeline 2), per set a high-priority HIP stream for the
stages that cannot use the chip's bandwidth (log-mel, DTW) and a low-priority
one for the HBM-bound ones (cost, log-prob gather), dependencies as events.
The single-batch-in-flight time is reported in the same line, and the
per-stage times / roofline are measured in that single-stream pass (events on
the stream the kernels are launched on).

Timing: the region of EXACTLY --steps steps (barrier + synchronize on both sides,
max over ranks) is repeated until at least --min-seconds of GPU work have been
timed (never fewer than 5 regions); ms_per_step / value are the MEDIAN region,
min and max are reported next to it.

The same line carries the transcribe()-level numbers ("e2e"): the batched second
pass of the naive strategy (whisper_timestamped/batched.py) and the default
strategy with 1 / 32 / 128 decoder streams, each next to the reference-shaped
CPU path (oracle/, same model on the CPU) and with per-word parity checks.

Files: this one is the argument parser, the orchestrator (which never touches
the GPU) and the process plumbing; the legs are benchlib/{kernel,cpu,second_pass,
default_strategy,recordings}_leg.py, one per child-process role; the synthetic
workloads and the in-leg oracle check are tests/workloads.py.
`python bench.py --gpus N` outside a launcher re-executes itself under
torch.distributed.run (one rank per GPU).  A parity check that does not hold
ends the run with a non-zero exit status (the line is still printed).

Prints ONE JSON line (rank 0).  metric = audio-seconds aligned per second.
"""
import argparse
import json
import os
import sys

import torch

from benchlib.common import METRIC, ROOT, json_scalar, log         # (also puts the package and tests/ on sys.path)

WORKLOAD_NAMES = ["kfull", "kfull256", "kreal", "largev3_fp16"]   # tests/workloads.py WORKLOADS


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="kfull", choices=WORKLOAD_NAMES + ["e2e_base32"],
                    help="kfull (default; its line also carries the transcribe()-level leg), the secondary kernel-level workloads, "
                         "or e2e_base32 = kfull with the e2e leg forced on")
    ap.add_argument("--min-seconds", type=float, default=1.0,
                    help="the --steps region is repeated until this much time has been measured (>= 5 regions)")
    ap.add_argument("--repeats", type=int, default=0, help="fixed number of timed regions (0 = from --min-seconds)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--e2e", default="auto", choices=["auto", "on", "off"],
                    help="transcribe()-level legs (whisper-base, batched.py); auto = with the default workload at N=1")
    ap.add_argument("--other-configs", default="on", choices=["on", "off"],
                    help="N=1, default workload: also run the kreal / kfull256 / largev3_fp16 kernel-level legs (other_configs)")
    ap.add_argument("--e2e-steps", type=int, default=6, help="launch sets of 32 chunks in the e2e timed region")
    ap.add_argument("--e2e-model", default="base", help="shapes of the e2e leg's model (whisper_double names: base = the "
                                                         "BASELINE config; small, medium, large-v3 ... for other shapes)")
    ap.add_argument("--e2e-windows", type=int, default=32, help="30 s chunks per launch set of the e2e leg")
    ap.add_argument("--e2e-cpu-budget", type=float, default=15.0)
    ap.add_argument("--gather-every", type=int, default=8,
                    help="N>1: result records of this many steps travel to rank 0 in one RCCL gather")
    ap.add_argument("--pipeline", type=int, default=2,
                    help="buffer sets in flight (HotPathPipeline depth): step k runs on stream set k %% N with its own output "
                         "buffers (the inputs are shared).  The line also carries the single-batch-in-flight time; per-stage "
                         "times and the roofline always come from the single-stream pass")
    ap.add_argument("--schedule", default="auto", choices=["serial", "hilo", "auto"],
                    help="whisper_timestamped.pipeline: serial = one stream per buffer set; hilo = per set a high-priority stream "
                         "(stft_mel, dtw_kernel) and a low-priority one (cost, log-prob); auto = the package's rule "
                         "(pipeline.choose_schedule)")
    ap.add_argument("--sub-batches", type=int, default=0,
                    help="launch a step's batch as this many chunk ranges, round-robin over the stream sets (0 = auto)")
    ap.add_argument("--align", default="auto", choices=["auto", "split", "fused"],
                    help="split: wt_cost_batch then wt_dtw_batch (two timed stages, batched kernels only); fused: ONE "
                         "wt_align_batch_v3 (small units through the fused kernel; timed as the cost stage); auto = fused for "
                         "the workloads that have small units (kreal)")
    # --- process plumbing (see orchestrate()): the measuring legs run in child processes of this script
    ap.add_argument("--role", default="orchestrate", choices=["orchestrate", "kernel", "cpu", "e2e"], help=argparse.SUPPRESS)
    ap.add_argument("--leg", default="fp32", choices=["fp32", "fp16", "efficient", "recordings"], help=argparse.SUPPRESS)
    ap.add_argument("--e2e-streams", type=int, default=32,
                    help="recordings per decoder op of the default-strategy leg (transcribe_batch; 32 = BASELINE configs[1])")
    ap.add_argument("--out", default=None, help=argparse.SUPPRESS)
    ap.add_argument("--secondary", action="store_true", help=argparse.SUPPRESS)     # a kernel leg of another BASELINE config
    ap.add_argument("--dry-run", action="store_true",
                    help="launcher / process-group plumbing only (gloo on the CPU, no kernels, meaningless numbers): what "
                         "tests/test_bench_launcher.py runs where there is no GPU")
    ap.add_argument("--inject-fault", default="", help=argparse.SUPPRESS)   # tests: "kernel", "e2e_fp16" ... abort that child
    args = ap.parse_args(argv)
    if args.workload == "e2e_base32":
        args.workload, args.e2e = "kfull", "on"
    return args


def role_e2e(args):
    from benchlib.common import make_emitter
    emit = make_emitter(args.out)
    if args.inject_fault == "e2e_" + args.leg:
        emit({"marker": "about to abort"})
        os.abort()
    if args.leg == "recordings":
        from benchlib.recordings_leg import role_recordings
        return role_recordings(args)
    torch.cuda.set_device(0)
    if args.leg == "efficient":
        from benchlib.default_strategy_leg import run_efficient_leg
        run_efficient_leg(args, emit)
    else:
        from benchlib.second_pass_leg import run_e2e
        run_e2e(torch.device("cuda", 0), args, args.leg, emit)


def run_child(role, extra, timeout_s, env=None):
    """One measuring leg in a child process of this script.  Returns (result dict or None, error string or None):
    the result is whatever the child published before it ended, however it ended."""
    import subprocess
    import tempfile
    fd, path = tempfile.mkstemp(prefix=f"wt_bench_{role}_", suffix=".json")
    os.close(fd)
    os.unlink(path)
    cmd = [sys.executable, os.path.abspath(__file__)] + sys.argv[1:] + ["--role", role, "--out", path] + list(extra)
    err = None
    # its own session: a leg that has to be stopped takes its own children (worker processes) with it
    env = dict(os.environ if env is None else env)
    env.setdefault("WT_BENCH_DUMP_STACKS_AFTER", str(max(5, timeout_s - 15)))
    proc = subprocess.Popen(cmd, stdout=sys.stderr, env=env, start_new_session=True)      # children never write to stdout
    try:
        rc = proc.wait(timeout=timeout_s)
        if rc < 0:
            err = f"signal {-rc}"
        elif rc != 0:
            err = f"exit {rc}"
    except subprocess.TimeoutExpired:
        err = f"timeout after {timeout_s} s"
        import signal
        try:
            os.killpg(proc.pid, signal.SIGKILL)
        except OSError:
            pass
        proc.wait()
    res = None
    if os.path.exists(path):
        try:
            res = json.load(open(path))
        except Exception:                       # noqa: BLE001
            res = None
        os.unlink(path)
    return res, err


def self_launch(args):
    """`python bench.py --gpus N` outside a launcher: re-exec under torch.distributed.run, one rank per GPU."""
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    if not args.dry_run and torch.cuda.device_count() < args.gpus:
        raise SystemExit(f"bench.py --gpus {args.gpus}: only {torch.cuda.device_count()} GPU(s) visible (one rank per GPU: RCCL "
                         f"refuses two ranks on one device)")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    sys.stdout.flush()
    os.execv(sys.executable, cmd)


def orchestrate(args):
    """The process the driver starts (one per rank).  It never touches the GPU itself: the kernel-level measurement,
    the CPU baselines and each transcribe()-level leg run in child processes that publish their results as they go,
    so a GPU fault in any leg costs that leg (reported as {"error": ...}), not the line.  stdout carries exactly ONE
    JSON line (rank 0), printed when every leg has ended; the headline is also logged to stderr as soon as the
    kernel leg has produced it."""
    world_env = os.environ.get("WORLD_SIZE")
    if args.gpus > 1 and world_env is None:
        self_launch(args)                                  # does not return
    rank = int(os.environ.get("RANK", "0"))
    world = int(world_env or "1")
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    inject = ["--inject-fault", ""]                        # a retried leg is not aborted again
    out, err = run_child("kernel", [], 420)
    attempts = 1
    if (out is None or err) and world == 1:
        # A leg that died is run once more and the line says so: the number is reported, the fault is not hidden.
        first = {"error": err, "published_before_the_fault": sorted(out) if out else None}
        out2, err2 = run_child("kernel", inject, 420)
        attempts = 2
        if out2 is not None and (out is None or not err2):
            out, err = out2, err2
        out = out or {}
        out["kernel_leg_first_attempt"] = first
    multi = None
    if world > 1 and args.e2e != "off":
        # BASELINE configs[3] on N GPUs: recordings across the ranks, decoder streams within a rank (every rank runs its child)
        multi, merr = run_child("e2e", ["--leg", "recordings"], 300)
        multi = multi or {}
        if merr:
            multi["error"] = merr
    if rank != 0:
        sys.exit(1 if err else 0)
    if world > 1 and out is not None:
        out["cpu_baseline"] = "N=1 line only"
        if multi is not None:
            out["transcribe_recordings"] = multi
    if out is None:
        out = {"metric": METRIC, "value": None, "unit": "audio-seconds/s",
               "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "higher_is_better": True}
    if err:
        out["kernel_leg_error"] = err
    out["kernel_leg_attempts"] = attempts
    # A recurrence of a GPU fault must not read as a clean pass: whatever the retried number is, the line says at its
    # top level that an attempt ended on a signal / non-zero exit.
    out["faulted"] = bool(attempts > 1 or err)
    print("[bench] kernel-level headline: " + json.dumps({k: out.get(k) for k in ("value", "unit", "ms_per_step", "roofline")}),
          file=sys.stderr, flush=True)
    if world == 1 and not args.dry_run and args.workload == "kfull" and args.other_configs == "on":
        # the other single-GPU BASELINE configurations, each as its own kernel-level leg (same child role, same timed
        # region, its own in-leg parity check): kreal = the reference's default per-segment call shape, kfull256 /
        # largev3_fp16 = BASELINE configs[4] shapes at N = 1 (256 chunks; large-v3 heads / mels / vocabulary, fp16 rows)
        others = {}
        for wl in ("kreal", "kfull256", "largev3_fp16"):
            leg, lerr = run_child("kernel", ["--workload", wl, "--min-seconds", "0.5", "--secondary"], 240)
            leg = leg or {}
            keep = {k: leg.get(k) for k in ("value", "unit", "ms_per_step", "timing", "single_batch_in_flight", "roofline",
                                            "stages", "parity_in_leg") if k in leg}
            if "config" in leg:
                keep["workload"] = leg["config"]["workload"]
                keep["alignment_entry"] = leg["config"]["alignment_entry"]
                keep["units_per_step"] = leg["config"].get("units_per_step")
            if lerr:
                keep["error"] = lerr
            others[wl] = keep
        out["other_configs"] = others
    if world == 1 and not args.dry_run:
        fixed_shape = not (args.workload == "kreal")
        if not args.no_cpu_baseline and fixed_shape:       # rank 0 at N=1 only (fixed-shape workloads)
            cpu, cerr = run_child("cpu", [], 300)
            out.update(cpu or {})
            if cerr:
                out["cpu_baseline_error"] = cerr
        if args.e2e == "on" or (args.e2e == "auto" and args.workload == "kfull"):
            e2e, e1 = run_child("e2e", ["--leg", "fp32"], 420)
            e2e = e2e or {}
            if e1:
                e2e["error"] = e1
            half, e2 = run_child("e2e", ["--leg", "fp16"], 420)
            half = half or {}
            half.pop("marker", None)
            e2e.update(half)
            if e2:
                e2e["fp16_legs_error"] = e2
            eff, e3 = run_child("e2e", ["--leg", "efficient"], 600)
            eff = eff or {}
            eff.pop("marker", None)
            if e3:
                eff["error"] = e3
            e2e["default_strategy"] = eff
            out["e2e_parity_failures"] = eff.get("parity_failures")
            # BASELINE configs[2]: whisper-small shapes through the same second pass (what the reference's beam-search
            # path re-runs, teacher forced: transcribe.py:1197-1262), 32 chunks per launch set
            small, e4 = run_child("e2e", ["--leg", "fp32", "--e2e-model", "small", "--no-cpu-baseline", "--e2e-steps", "3"], 300)
            small = small or {}
            if e4:
                small["error"] = e4
            half_s, e5 = run_child("e2e", ["--leg", "fp16", "--e2e-model", "small", "--e2e-steps", "3"], 300)
            if half_s:
                half_s.pop("marker", None)
                small.update(half_s)
            if e5:
                small["fp16_legs_error"] = e5
            e2e["whisper_small_shapes"] = small
            out["e2e"] = e2e
            # the second half of BASELINE.json's metric ("...; max |dt_word| vs ref"), from the legs that compare words with
            # the reference-shaped CPU path: the teacher-forced second pass and the default strategy
            try:
                dts = [(e2e.get("parity_vs_cpu_reference_path") or {}).get("max_abs_dt_word_s"),
                       (eff.get("parity_vs_cpu_reference_path") or {}).get("max_abs_dt_word_s")]
                dts = [float(x) for x in dts if x is not None]
                out["max_abs_dt_word_vs_ref_s"] = max(dts) if dts else None
            except Exception:                              # noqa: BLE001 -- a summary key must never cost the line
                out["max_abs_dt_word_vs_ref_s"] = None
    # A parity check that did not hold fails the run (the line is still printed: the failure is named in it).
    failed = [f for f in (out.get("e2e_parity_failures") or [])]
    kernel_parity = out.get("parity_in_leg")
    if kernel_parity is not None and not kernel_parity.get("ok", False):
        failed.append({"leg": "kernel", "detail": kernel_parity})
    for name, leg in (out.get("other_configs") or {}).items():
        if leg.get("parity_in_leg") is not None and not leg["parity_in_leg"].get("ok", False):
            failed.append({"leg": name, "detail": leg["parity_in_leg"]})
    out["parity_failures"] = failed
    print(json.dumps(out, default=json_scalar), flush=True)
    if out.get("value") is None:
        sys.exit(1)
    if failed:
        log(f"{len(failed)} parity check(s) did not hold: exit status 3")
        sys.exit(3)


def cap_threads_per_rank():
    """N ranks on one host: each gets its share of the cores for torch's intra-op pool (and its children inherit it) --
    eight ranks with every core each is what made the default-strategy CPU leg 12x slower at 128 threads than at 16."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1:
        n = max(1, (os.cpu_count() or world) // world)
        torch.set_num_threads(n)
        os.environ["OMP_NUM_THREADS"] = str(n)
    return torch.get_num_threads()


def main():
    # multi-process GPU work on this pool's hosts (dmabuf IPC only): without it RCCL's communicator set-up fails with
    # hipIpcGetMemHandle: invalid argument.  Before the first HIP call; the children and self-launched ranks inherit it.
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    args = parse_args()
    cap_threads_per_rank()
    if args.role == "orchestrate":
        return orchestrate(args)
    # children: stdout belongs to the parent's single JSON line -- everything libraries print goes to stderr
    sys.stdout.flush()
    os.dup2(2, 1)
    # a leg that is still running shortly before its parent gives up on it says where (every thread's Python stack)
    dump_after = float(os.environ.get("WT_BENCH_DUMP_STACKS_AFTER", "0") or 0)
    if dump_after > 0:
        import faulthandler
        faulthandler.dump_traceback_later(dump_after, exit=False, file=sys.stderr)
    if args.role == "kernel":
        from benchlib.kernel_leg import role_kernel
        role_kernel(args)
    elif args.role == "cpu":
        from benchlib.cpu_leg import role_cpu
        role_cpu(args)
    else:
        role_e2e(args)


if __name__ == "__main__":
    main()
