"""Small helpers with the names openai-whisper exposes in whisper.utils."""
import zlib


def exact_div(x, y):
    assert x % y == 0
    return x // y


def compression_ratio(text) -> float:
    raw = text.encode("utf-8")
    return len(raw) / len(zlib.compress(raw))


def format_timestamp(seconds: float, always_include_hours: bool = False, decimal_marker: str = "."):
    assert seconds >= 0, "non-negative timestamp expected"
    ms = round(seconds * 1000.0)
    hours, ms = divmod(ms, 3_600_000)
    minutes, ms = divmod(ms, 60_000)
    secs, ms = divmod(ms, 1_000)
    head = f"{hours:02d}:" if always_include_hours or hours > 0 else ""
    return f"{head}{minutes:02d}:{secs:02d}{decimal_marker}{ms:03d}"


# --- what whisper_timestamped's command line takes from whisper.utils (argparse helpers, result writers) ---------------
def str2bool(string):
    table = {"True": True, "False": False}
    if string in table:
        return table[string]
    raise ValueError(f"Expected one of {set(table)}, got {string}")


def optional_int(string):
    return None if string == "None" else int(string)


def optional_float(string):
    return None if string == "None" else float(string)


class _Writer:
    ext = ""

    def __init__(self, output_dir):
        self.output_dir = output_dir

    def write_result(self, result, file, options=None):
        raise NotImplementedError


class WriteTXT(_Writer):
    ext = "txt"

    def write_result(self, result, file, options=None):
        for segment in result["segments"]:
            print(segment["text"].strip(), file=file, flush=True)


class WriteVTT(_Writer):
    ext = "vtt"

    def write_result(self, result, file, options=None):
        print("WEBVTT\n", file=file)
        for s in result["segments"]:
            print(f"{format_timestamp(s['start'])} --> {format_timestamp(s['end'])}\n{s['text'].strip().replace('-->', '->')}\n",
                  file=file, flush=True)


class WriteSRT(_Writer):
    ext = "srt"

    def write_result(self, result, file, options=None):
        for i, s in enumerate(result["segments"], start=1):
            a = format_timestamp(s["start"], always_include_hours=True, decimal_marker=",")
            b = format_timestamp(s["end"], always_include_hours=True, decimal_marker=",")
            print(f"{i}\n{a} --> {b}\n{s['text'].strip().replace('-->', '->')}\n", file=file, flush=True)


class WriteTSV(_Writer):
    ext = "tsv"

    def write_result(self, result, file, options=None):
        print("start", "end", "text", sep="\t", file=file)
        for s in result["segments"]:
            print(round(1000 * s["start"]), round(1000 * s["end"]), s["text"].strip().replace("\t", " "), sep="\t", file=file, flush=True)


def get_writer(output_format, output_dir):
    return {"txt": WriteTXT, "vtt": WriteVTT, "srt": WriteSRT, "tsv": WriteTSV}[output_format](output_dir)
