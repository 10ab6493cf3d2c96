"""Packs the reference's ten demo recordings (Data/voice/records/audio/segment_ph*.wav: mono s16, 44.1 kHz, 2.86-37.2 s) as DATA
into tests/golden/demo_full.npz so that BASELINE.json configs[4] (the pipeline end to end on the full Data/voice set) can run on the
GPU box, where /root/reference does not exist.  Samples and the rate only; run in the build container:

    python tests/golden/make_goldens_demo_full.py
"""
import glob
import os
import re
import wave

import numpy as np

SRC = "/root/reference/Data/voice/records/audio"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "demo_full.npz")


def main():
    clips, rate = {}, None
    for p in sorted(glob.glob(os.path.join(SRC, "segment_ph*.wav")), key=lambda q: int(re.search(r"(\d+)", os.path.basename(q)).group(1))):
        with wave.open(p, "rb") as w:
            assert w.getnchannels() == 1 and w.getsampwidth() == 2
            rate = w.getframerate() if rate is None else rate
            assert rate == w.getframerate()
            clips[os.path.basename(p)[:-4]] = np.frombuffer(w.readframes(w.getnframes()), dtype="<i2").copy()
    np.savez_compressed(OUT, rate=np.int32(rate), **clips)
    print(OUT, {k: round(len(v) / rate, 2) for k, v in clips.items()}, os.path.getsize(OUT))


if __name__ == "__main__":
    main()
