"""Device time of the probabilistic YIN on 64 x 10 s clips at 16 kHz (HIP events via the engine's profiler)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import prosody_control_french_tts_amd as pkg
from prosody_control_french_tts_amd import synth
from prosody_control_french_tts_amd.visualisation import acoustic_analysis as AA

n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
clips = synth.synth_batch(n, 10.0, 16000, first=0)
eng = pkg.ProsodyEngine(0)
eng.upload(clips, 16000)
plan, tables, _ = AA.pyin_plan(16000)
eng.pyin_run(plan, tables)
eng.profile_enable(True); eng.profile_reset()
for _ in range(3):
    eng.pyin_run(plan, tables)
eng.sync()
p = eng.profile()
for k in ("k_pyin_frames", "k_pyin_viterbi"):
    print(k, f"{p[k]['total_ms'] / p[k]['launches']:.2f} ms per launch ({n} clips x 10 s, {n * 626} frames)")
res = AA.pyin_batch(eng)
print("voiced fraction", float(np.mean(np.concatenate([r[1] for r in res]))), "median f0", float(np.nanmedian(np.concatenate([r[0] for r in res]))))
eng.close()
