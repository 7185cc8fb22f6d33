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
