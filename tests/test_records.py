"""whisper_timestamped/records.py: transcribe() dictionaries as byte records (what the ranks send to rank 0)."""
import json
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))


def _goldens():
    cases = json.load(open(os.path.join(HERE, "golden", "transcribe_cases.json")))
    return [c["expected"] for c in cases]


def test_messages_of_several_ranks_make_one_table_of_lazily_decoded_results():
    from whisper_timestamped import records as R
    pool = _goldens()
    pool[1] = dict(pool[1], language_probs={"en": 0.9}, speech_activity=[{"start": 0.0, "end": 1.5}])     # any keys travel
    per_rank = [[(7, pool[0]), (2, pool[3])], [], [(0, pool[5])], [(5, pool[1]), (1, pool[2]), (9, pool[4])]]
    messages = []
    for pairs in per_rank:
        msg = R.pack_many(pairs)
        assert msg.dtype == np.int32 and int(msg[0]) == len(pairs)
        messages.append(np.concatenate([msg, np.full(17, -1, np.int32)]))      # (padding behind a message is ignored)
    table = R.table_of(messages)
    assert len(table) == 6 and sorted(table.indices) == [0, 1, 2, 5, 7, 9]
    want = {i: r for pairs in per_rank for i, r in pairs}
    for i, r in want.items():
        got = table.dict(i)
        assert got == r and json.dumps(got, ensure_ascii=False) == json.dumps(r, ensure_ascii=False)      # key order and int-ness too
        assert table.nbytes(i) > 100
    assert table.dicts() == [want[i] for i in sorted(want)]
    empty = R.table_of([R.pack_many([])])
    assert len(empty) == 0 and empty.dicts() == []
