import sys; sys.path.insert(0,'.')
import numpy as np
from oracle import oracle as O
from prosody_control_french_tts_amd import engine as E, synth
import prosody_control_french_tts_amd as P
clips=[synth.synth_clip(i, seconds=3.0) for i in range(4)]
eng=P.ProsodyEngine(0); eng.upload(clips,16000)
for floor in (75.0,150.0):
    sl=eng.whole_clip_slices()
    res=eng.pitch(sl, E.PitchParams.praat(floor,600.0), want_strength=True)
    off=res['frame_offsets']
    for k in range(len(clips)):
        want=O.pitch_ac(clips[k]/32768.0,1/16000,0.5/16000,O.praat_params(floor,600.0))
        f0=res['f0'][off[k]:off[k+1]]; v=want['f0']>0
        rel=np.abs(f0[v]-want['f0'][v])/want['f0'][v]
        print(floor,k,'same vuv',np.array_equal(f0>0,v),'max rel',rel.max(),'mean rel',rel.mean(),'strength diff',np.abs(res['strength'][off[k]:off[k+1]][v]-want['strength'][v]).max(), 'meanlog diff', abs(res['summary'][k]['mean_log_f0']-np.mean(np.log(want['f0'][v]))))
