"""The JSON / CSV surface around transcribe()'s result: which keys survive, how floats
are rounded, how segments and words become table rows.  Mirrors filtered_keys / flatten / remove_keys / write_csv of
/root/reference/whisper_timestamped/transcribe.py:2299-2325,3185-3201 (same names, arguments and results); the
command-line shell and the subtitle writers around them stay out of scope."""
import csv

KEPT_KEYS = ("text", "segments", "words", "language", "start", "end", "confidence", "language_probs", "speech_activity")


def filtered_keys(result, keys=KEPT_KEYS):
    """The compact view: only `keys` survive at every depth, floats are rounded to 2 decimals, the `language_probs`
    table is kept whole (transcribe.py:3185-3201)."""
    if isinstance(result, dict):
        out = {}
        for name, value in result.items():
            if name in keys:
                out[name] = value if name == "language_probs" else filtered_keys(value, keys)
        return out
    if isinstance(result, list):
        return [filtered_keys(item, keys) for item in result]
    if isinstance(result, float):
        return round(result, 2)
    return result


def flatten(list_of_lists, key=None):
    """Items of every sub-list; with `key`, items of every element's `key` list (missing key = nothing):
    flatten(result["segments"], "words") walks all words (transcribe.py:2299-2302)."""
    for sub in list_of_lists:
        yield from (sub.get(key, []) if key else sub)


def remove_keys(list_of_dicts, key):
    """Every dictionary without `key` (transcribe.py:2304-2306; key order is not part of the contract)."""
    for d in list_of_dicts:
        yield {k: v for k, v in d.items() if k != key}


def write_csv(transcript, file, sep=",", text_first=True, format_timestamps=None, header=False):
    """One row per element of `transcript` (segments or words): stripped text, start, end -- or start, end, text
    (transcribe.py:2309-2325).  header: True = the default column names, a list = those names, falsy = none."""
    fmt = format_timestamps if format_timestamps is not None else (lambda t: t)
    out = csv.writer(file, delimiter=sep)
    if header is True:
        header = ["text", "start", "end"] if text_first else ["start", "end", "text"]
    if header:
        out.writerow(header)
    for item in transcript:
        text, times = item["text"].strip(), [fmt(item["start"]), fmt(item["end"])]
        out.writerow([text] + times if text_first else times + [text])
