"""Command line of whisper_timestamped: same options, defaults and output files as the reference's ``cli()``
(/root/reference/whisper_timestamped/transcribe.py:2964-3182), so that its goldens under tests/expected/ can be
replayed with ``--model <checkpoint>`` wherever trained weights exist.  Host glue only: it parses, calls
``transcribe_timestamped`` (whose alignment runs on the MI355X) and writes files through the backend's own writers.
"""
from __future__ import annotations

import json
import logging
import os
import sys

import numpy as np
import torch

from . import backend as _backend
from .output import filtered_keys, flatten, remove_keys, write_csv
from .transcribe import load_model, transcribe_batch, transcribe_timestamped

logger = logging.getLogger("whisper_timestamped")
VALID_FORMATS = ["txt", "vtt", "srt", "tsv", "csv", "json"]


def _writers():
    """txt / srt / vtt / tsv writers of whichever openai-whisper is installed (transcribe.py:2973-2999)."""
    utils = _backend.whisper().utils
    if hasattr(utils, "write_txt"):                       # before openai-whisper da600abd
        return dict(txt=utils.write_txt, srt=utils.write_srt, vtt=utils.write_vtt,
                    tsv=lambda transcript, file: write_csv(transcript, file, sep="\t", header=True, text_first=False,
                                                           format_timestamps=lambda x: round(1000 * x)))

    def writer_for(fmt):
        def write(transcript, file):
            w = utils.get_writer(fmt, os.path.curdir)
            try:
                return w.write_result({"segments": list(transcript)}, file,
                                      {"highlight_words": False, "max_line_width": None, "max_line_count": None})
            except TypeError:                                 # openai-whisper <= 20230314
                return w.write_result({"segments": transcript}, file)
        return write
    return {fmt: writer_for(fmt) for fmt in ("txt", "srt", "vtt", "tsv")}


def _output_formats(string):
    if string == "all":
        return VALID_FORMATS
    formats = string.split(",")
    for fmt in formats:
        if fmt not in VALID_FORMATS:
            raise ValueError(f"Expected one of {VALID_FORMATS}, got {fmt}")
    return formats


def build_parser():
    import argparse
    from . import __version__
    w = _backend.whisper()
    utils = w.utils
    str2bool, optional_float, optional_int = utils.str2bool, utils.optional_float, utils.optional_int
    p = argparse.ArgumentParser(description="Transcribe a single audio with whisper and compute word timestamps",
                                formatter_class=argparse.ArgumentDefaultsHelpFormatter)
    p.add_argument("-v", "--version", help="show version and exit", action="version", version=f"{__version__}")
    p.add_argument("--versions", help="show versions (of whisper-timestamped and whisper) and exit", action="version",
                   version=f"{__version__} -- Whisper {getattr(w, '__version__', '?')} in "
                           f"{os.path.realpath(os.path.dirname(w.__file__))}")
    p.add_argument("audio", help="audio file(s) to transcribe", nargs="+")
    p.add_argument("--model", help=f"name of the Whisper model to use. Examples: {', '.join(w.available_models())}", default="small")
    p.add_argument("--model_dir", default=None, type=str, help="the path to save model files; uses ~/.cache/whisper by default")
    p.add_argument("--device", default="cuda:0" if torch.cuda.is_available() else "cpu", help="device to use for PyTorch inference")
    p.add_argument("--backend", default="openai-whisper", choices=["openai-whisper", "transformers"], type=str, help="Which backend to use")
    p.add_argument("--output_dir", "-o", default=None, type=str, help="directory to save the outputs")
    p.add_argument("--output_format", "-f", default="all", type=_output_formats,
                   help=f"Format(s) of the output file(s). Possible formats are: {', '.join(VALID_FORMATS)}. Several formats can be "
                        "specified by using commas (ex: \"json,vtt,srt\"). By default (\"all\"), all available formats will be produced")
    p.add_argument("--task", default="transcribe", choices=["transcribe", "translate"], type=str,
                   help="whether to perform X->X speech recognition ('transcribe') or X->English translation ('translate')")
    p.add_argument("--language", default=None, help="language spoken in the audio, specify None to perform language detection.",
                   choices=sorted(w.tokenizer.LANGUAGES.keys()) + sorted(k.title() for k in w.tokenizer.TO_LANGUAGE_CODE.keys()))
    p.add_argument("--vad", default=False,
                   help="whether to run Voice Activity Detection (VAD) to remove non-speech segment before applying Whisper model "
                        "(removes hallucinations). Can be: True, False, auditok, silero (default when vad=True), silero:3.1 (or "
                        "another version), or a list of timestamps in seconds (e.g. \"[(0.0, 3.50), (32.43, 36.43)]\").")
    p.add_argument("--detect_disfluencies", default=False, type=str2bool, help="whether to try to detect disfluencies, marking them as special words [*]")
    p.add_argument("--recompute_all_timestamps", default=False, type=str2bool, help="Do not rely at all on Whisper timestamps (experimental)")
    p.add_argument("--punctuations_with_words", default=True, type=str2bool, help="whether to include punctuations in the words")
    p.add_argument("--temperature", default=0.0, type=float, help="temperature to use for sampling")
    p.add_argument("--best_of", type=optional_int, default=None, help="number of candidates when sampling with non-zero temperature")
    p.add_argument("--beam_size", type=optional_int, default=None, help="number of beams in beam search, only applicable when temperature is zero")
    p.add_argument("--patience", type=float, default=None, help="optional patience value to use in beam decoding")
    p.add_argument("--length_penalty", type=float, default=None, help="optional token length penalty coefficient (alpha)")
    p.add_argument("--suppress_tokens", default="-1", type=str, help="comma-separated list of token ids to suppress during sampling")
    p.add_argument("--initial_prompt", default=None, type=str, help="optional text to provide as a prompt for the first window.")
    p.add_argument("--condition_on_previous_text", default=True, type=str2bool, help="provide the previous output of the model as a prompt for the next window")
    p.add_argument("--fp16", default=None, type=str2bool, help="whether to perform inference in fp16; automatic by default")
    p.add_argument("--temperature_increment_on_fallback", default=0.0, type=optional_float, help="temperature to increase when falling back")
    p.add_argument("--compression_ratio_threshold", default=2.4, type=optional_float, help="gzip compression ratio above which decoding is treated as failed")
    p.add_argument("--logprob_threshold", default=-1.0, type=optional_float, help="average log probability below which decoding is treated as failed")
    p.add_argument("--no_speech_threshold", default=0.6, type=optional_float, help="<|nospeech|> probability above which a failed window is silence")
    p.add_argument("--threads", default=0, type=optional_int, help="number of threads used by torch for CPU inference")
    p.add_argument("--compute_confidence", default=True, type=str2bool, help="whether to compute confidence scores for words")
    p.add_argument("--verbose", type=str2bool, default=False, help="whether to print out the progress and debug messages of Whisper")
    p.add_argument("--plot", default=False, action="store_true", help="plot word alignments")
    p.add_argument("--debug", default=False, action="store_true", help="print some debug information about word alignment")

    def shortcut(**values):
        class Set(argparse.Action):
            def __init__(self, option_strings, dest, nargs=None, **kw):
                assert nargs is None
                super().__init__(option_strings, dest, nargs=0, **kw)

            def __call__(self, parser, namespace, vals, option_string=None):
                for k, v in values.items():
                    setattr(namespace, k, v)
        return Set
    p.add_argument("--accurate", action=shortcut(best_of=5, beam_size=5, temperature_increment_on_fallback=0.2),
                   help="Shortcut to use the same default option as in openai-whisper (best_of=5, beam_search=5, temperature_increment_on_fallback=0.2)")
    p.add_argument("--efficient", action=shortcut(best_of=None, beam_size=None, temperature_increment_on_fallback=None),
                   help="Shortcut to disable beam size and options that requires to sample several times, for an efficient decoding")
    p.add_argument("--streams", type=int, default=0,
                   help="(not in the reference) with several audio files: step up to this many of them through the decoder "
                        "together (transcribe_batch); 0 = one file after the other, as the reference does.  Same output files")
    p.add_argument("--naive", default=False, action="store_true",
                   help="use naive approach, doing inference twice (once to get the transcription, once to get word timestamps and confidence scores).")
    return p


def cli(argv=None):
    args = build_parser().parse_args(argv).__dict__
    args.pop("accurate")
    args.pop("efficient")
    temperature = args.pop("temperature")
    increment = args.pop("temperature_increment_on_fallback")
    temperature = tuple(np.arange(temperature, 1.0 + 1e-6, increment)) if increment else [temperature]
    threads = args.pop("threads")
    if threads:
        torch.set_num_threads(threads)
    audio_files = args.pop("audio")
    model = load_model(args.pop("model"), device=args.pop("device"), download_root=args.pop("model_dir"),
                       backend=args.pop("backend"))
    output_format = args.pop("output_format")
    plot = args.pop("plot")
    logging.basicConfig()
    if args.pop("debug"):
        logger.setLevel(logging.DEBUG)
        logging.getLogger("WHISPER").setLevel(logging.DEBUG)
    output_dir = args.pop("output_dir")
    if output_dir and not os.path.isdir(output_dir):
        os.makedirs(output_dir)
    args["naive_approach"] = args.pop("naive")
    args["remove_punctuation_from_words"] = not args.pop("punctuations_with_words")
    args["compute_word_confidence"] = args.pop("compute_confidence")
    args["trust_whisper_timestamps"] = not args.pop("recompute_all_timestamps")
    write = _writers()
    n_streams = args.pop("streams")
    results = None
    if n_streams and n_streams > 1 and len(audio_files) > 1 and not plot:
        # independent recordings: up to n_streams of them per decoder op (calls the B-stream path cannot take -- beam
        # search, temperature fallback, vad -- are decoded one after the other by transcribe_batch itself)
        results = transcribe_batch(model, audio_files, max_streams=n_streams, temperature=temperature, **args)

    for k, audio_path in enumerate(audio_files):
        outname = os.path.join(output_dir, os.path.basename(audio_path)) if output_dir else None
        result = results[k] if results is not None else \
            transcribe_timestamped(model, audio_path, temperature=temperature,
                                   plot_word_alignment=outname if (outname and plot) else plot, **args)
        if not output_dir:
            # (the B-stream path prints no segments while it decodes: its results are always dumped, verbose or not)
            if not args["verbose"] or results is not None:
                json.dump(filtered_keys(result), sys.stdout, indent=2, ensure_ascii=False)
            continue
        segments = result["segments"]

        def save(suffix, fn, rows):
            with open(outname + suffix, "w", encoding="utf-8") as f:
                fn(rows, file=f)
        if "json" in output_format:
            with open(outname + ".words.json", "w", encoding="utf-8") as f:
                json.dump(result, f, indent=2, ensure_ascii=False, default=float)
        if "txt" in output_format:
            save(".txt", write["txt"], segments)
        for fmt in ("vtt", "srt"):                  # segment-level file without the words, word-level file beside it
            if fmt in output_format:
                save("." + fmt, write[fmt], remove_keys(segments, "words"))
                save(".words." + fmt, write[fmt], flatten(segments, "words"))
        if "csv" in output_format:
            save(".csv", write_csv, segments)
            save(".words.csv", write_csv, flatten(segments, "words"))
        if "tsv" in output_format:
            save(".tsv", write["tsv"], segments)
            save(".words.tsv", write["tsv"], flatten(segments, "words"))


if __name__ == "__main__":
    cli()
