"""Device time of the HBM-bound framing kernels (k_energy: seven exact reductions per sample; k_frame_energy: one) at the C2 batch
(256 x 10 s = 82 MB, smaller than the 256 MB Infinity Cache: a repeated launch is served on-die) and at the C4 shard size
(1 250 x 10 s = 400 MB per GPU, beyond it: the HBM number).  HIP events via the engine's profiler.  usage: energy_rate.py [clips ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import prosody_control_french_tts_amd as pkg
from prosody_control_french_tts_amd import synth

sizes = [int(a) for a in sys.argv[1:]] or [256, 1250]
base = synth.synth_batch(256, 10.0, 16000, first=0)
eng = pkg.ProsodyEngine(0)
for n in sizes:
    clips = [base[i % 256] for i in range(n)]
    eng.upload(clips, 16000)
    sl = eng.whole_clip_slices()
    nbytes = sum(len(c) for c in clips) * 2
    for name, run, extra in (("k_energy", lambda: eng.energy_run(sl, 500), 0),
                             ("k_frame_energy", lambda: eng.frame_energy_run(800, 800, requantize=False), n * 200 * 12)):
        for _ in range(5):
            run()
        eng.profile_enable(True); eng.profile_reset()
        for _ in range(30):
            run()
        eng.sync()
        p = eng.profile()[name]
        ms = p["total_ms"] / p["launches"]
        eng.profile_enable(False)
        print(f"{n:5d} clips ({nbytes / 1e6:6.1f} MB)  {name:15s} {ms * 1e3:7.1f} us per launch  {(nbytes + extra) / ms / 1e6:7.0f} GB/s = {(nbytes + extra) / ms / 1e6 / 80:.1f} % of 8 TB/s")
eng.close()
