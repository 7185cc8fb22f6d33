#!/usr/bin/env python3
"""Golden for the island-sharded long-form job (BASELINE config 4's shape, scaled to CPU minutes).

Runs the REFERENCE's own transcribe_timestamped (/root/reference, unmodified; whisper double + C DTW stub as in
make_golden_transcribe.py) on the crop of every speech island of one synthetic recording and stores each island's
public JSON surface + the tokens its decoder sampled -> tests/golden/islands_job.json.  The sharded job
(whisper_timestamped.sharding.transcribe_islands) must reproduce these per island, whatever the number of ranks.

Usage (build container only; the reference does not travel):  python tests/golden/make_golden_islands.py
"""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "whisper-timestamped_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)
from golden import make_golden_transcribe as G  # noqa: E402

OUT = os.path.join(HERE, "islands_job.json")
ML, EOT_ML = 50364, 50257


def job():
    def seg(seed, s, n, e):
        return (s, G.text_ids(seed, n) if seed % 2 else [None] * n, e)
    W = G.window_script
    return dict(
        name="islands_job", model="tiny", model_seed=0, audio_s=78.0, audio_seed=41, opts=dict(language="en"),
        islands=[(0.5, 12.5), (15.0, 49.0), (52.25, 60.0), (61.0, 77.5)],
        scripts=[
            [W(ML, EOT_ML, [seg(11, 10, 7, 180), seg(12, 200, 9, 560)], "eot")],
            [W(ML, EOT_ML, [seg(13, 0, 8, 300), seg(14, 310, 6, 700), seg(15, 720, 10, 1300)], "pair"),
             W(ML, EOT_ML, [seg(16, 10, 5, 180)], "eot")],
            [W(ML, EOT_ML, [seg(17, 5, 6, 350)], "eot")],
            [W(ML, EOT_ML, [seg(18, 20, 8, 400), seg(19, 410, 6, 800)], "eot")],
        ])


def main():
    from whisper_double.decoding import Script, set_script
    ref = G.load_reference()
    j = job()
    model, audio, _ = G.build_case(dict(j, script=None))
    j["expected"], j["recorded"] = [], []
    for (s, e), windows in zip(j["islands"], j["scripts"]):
        crop = audio[int(round(s * 16000)):int(round(e * 16000))]
        script = set_script(Script(windows))
        try:
            result = ref.transcribe_timestamped(model, crop, fp16=False, **j["opts"])
        finally:
            set_script(None)
        view = json.loads(json.dumps(G.public_view(result), default=float))
        j["expected"].append(view)
        j["recorded"].append(script.record)
        print(f"island {s:6.2f}-{e:6.2f}: segments={len(view['segments'])} "
              f"words={sum(len(x['words']) for x in view['segments'])} windows={len(script.record)}")
    with open(OUT, "w", encoding="utf-8") as f:
        json.dump(j, f, ensure_ascii=False, indent=0)
    print("wrote", OUT)


if __name__ == "__main__":
    main()
