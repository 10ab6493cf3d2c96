"""BASELINE.json configs[4] on ONE GPU: the pipeline end to end on the full Data/voice set, SSML output diffed against the CPU path.

``run_all`` (the reference's ``__main__`` driver, Code/audioPipeline.py:1121-1166) runs the REAL steps ``"Align+Transcribe"`` ->
``"Measure & Build SSML"`` -> ``"Final Transcribe"`` (config.yaml:46-54, step table :1076-1103) on the reference's ten demo
recordings (``tests/golden/demo_full.npz``: 2.86-37.2 s at 44.1 kHz, so the 37.2 s one needs two Whisper windows), then the
break-prediction forward (``predict_breaks``).  Nothing is monkey-patched.  What cannot exist offline is stood in for by data, not by
code: a random-init miniature Whisper / BERT written as checkpoint files (no trained weights offline: the words are arbitrary, the
chain they drive is the real one) and a deterministic raw-synthesis stand-in derived from the natural audio (the "Raw Synthesis" step is
an Azure call: out of scope, skipped with a warning as in every run of this package).

The diff is a MEASUREMENT diff, not an independent SSML diff: the three CSVs of the step are rebuilt by the SAME ``SsmlTagger`` (product code on
both sides) fed by the CPU ORACLE's measurements over the TextGrids the aligner has just written (the reference's closures on the restated
Praat / pyloudnorm / pydub arithmetic with the reference's meters), and must be text-identical; F0 medians within 1e-6 relative, LUFS within
1e-6 LU, durations equal.  What it shows is that the GPU's measurements lead to the same tables as the CPU's; that the tagger turns measurements
into the reference's strings is pinned separately, on CPU, by golden G7 (the reference's own ``measure_prosody_and_build_ssml`` run with
scripted measurements: tests/test_goldens.py).  The Whisper / BERT checkpoints are random-init miniatures: the chain is real, the words are noise.  The 8-rank half of configs[4] needs a node
this pool does not have: the same chain runs at world size 2 under gloo in tests/test_c5_chain.py.
"""
import logging
import wave
from pathlib import Path

import numpy as np
import pytest

from prosody_control_french_tts_amd import audio_pipeline as AP
from prosody_control_french_tts_amd import bert_weights as BW
from prosody_control_french_tts_amd import engine as E
from prosody_control_french_tts_amd import tagger as T
from prosody_control_french_tts_amd import textgrid_io as TG

pytestmark = pytest.mark.gpu
GOLD = Path(__file__).parent / "golden"
VOICE = "records"                                   # Data/voice/records/audio/segment_ph*.wav in the reference's tree


# the multi-byte tokens of the miniature vocabulary: whole French words (leading space = a new word, as in Whisper's vocabulary), some with
# the punctuation the tagger reacts to (sentence ends, commas) and closed-class words the pause filter knows (Code/audioPipeline.py:27)
WORDS = [w.encode("utf-8") for w in (
    " bonjour", " le", " monde,", " voilà", " une", " phrase.", " très", " longue", " ici?", " oui", " de", " la", " mer!", " encore", " un", " mot",
    " nous", " allons", " parler", " doucement.", " et", " puis", " vite,", " mais", " pas", " trop", " les", " enfants", " jouent", " dehors.",
    " il", " fait", " beau", " aujourd'hui,", " elle", " chante", " souvent", " avec", " ses", " amis.", " quand", " vient", " le soir,", " tout", " devient",
    " calme", " dans", " la ville.")]


def write_wav(path, pcm, rate):
    path.parent.mkdir(parents=True, exist_ok=True)
    with wave.open(str(path), "wb") as w:
        w.setnchannels(1); w.setsampwidth(2); w.setframerate(rate); w.writeframes(np.asarray(pcm, dtype="<i2").tobytes())


def raw_synthesis_stand_in(nat: np.ndarray, k: int) -> np.ndarray:
    """What the (out of scope) "Raw Synthesis" step would leave in ``<voice>_raw/audio``: a 16 kHz file (Azure's RIFF default,
    Code/Preprocessing/get_synth.py:46-51).  Deterministic from the natural recording: polyphase 44.1 -> 16 kHz, a slightly faster
    delivery (a few percent of the frames dropped at the end), a different level."""
    from scipy.signal import resample_poly
    y = resample_poly(nat.astype(np.float64), 160, 441)
    y = y[: int(len(y) * (0.90 + 0.01 * (k % 7)))] * (0.55 + 0.05 * (k % 5))
    return np.clip(np.round(y), -32768, 32767).astype(np.int16)


def lay_out_voice(base: Path):
    z = np.load(GOLD / "demo_full.npz")
    rate = int(z["rate"])
    names = sorted((k for k in z.files if k != "rate"), key=T.segment_sort_key)
    pcm, rates = {}, {}
    for k, n in enumerate(names):
        nat = z[n]
        syn = raw_synthesis_stand_in(nat, k)
        write_wav(base / "Data" / "voice" / VOICE / "audio" / f"{n}.wav", nat, rate)
        write_wav(base / "Data" / "voice" / f"{VOICE}_raw" / "audio" / f"{n}.wav", syn, 16000)
        pcm[("nat", n)], pcm[("syn", n)] = nat, syn
        rates[("nat", n)], rates[("syn", n)] = rate, 16000
    # the (out of scope) "Synthesize+Merge" step's product, which "Final Transcribe" reads: 45 s of 16 kHz audio -> two windows
    out = np.concatenate([pcm[("syn", n)] for n in names[:3]])[: 45 * 16000]
    write_wav(base / "Out" / "results" / VOICE / "OUT.wav", out, 16000)
    return names, pcm, rates, rate


def test_c5_full_voice_end_to_end_and_ssml_diff_against_the_cpu_path(engine, tmp_path):
    from tests.test_gpu_aligner import write_model_dir
    from tests.test_gpu_pipeline import MixedRateOracle
    names, pcm, rates, nat_rate = lay_out_voice(tmp_path)
    assert len(names) == 10 and max(len(pcm[("nat", n)]) for n in names) > 30 * nat_rate
    model_root = tmp_path / "whisper_dir"
    write_model_dir(model_root, merges=WORDS, word_gain=3.0, eot_gain=0.3)
    cfg = {"data_dir": "Data/voice", "out_dir": "Out", "azure_voice_name": "fr-FR-HenriNeural", "whisper_device": "cuda:0", "whisper_model": "medium",
           "whisper_dir": str(model_root), "voice_names": [VOICE], "multiprocessing": True, "num_processes": 5,
           "prosody_settings": {"baseline_window": 10, "pitch_semitones": 1.3, "pitch_lower_clip_factor": 0.7, "pitch_offset_semitones": 5.0, "volume_pct": 10.0,
                                "rate_percent": 10.0, "threshold_duration_before_slowing_down": 1.0, "slow_floor_per_sec": 2.0, "smoothing_alpha": 0.2,
                                "max_jump_percent": 8, "end_punctuation_pause_ms": 500, "inter_syntagme_pause_factor": 1},      # config.yaml:25-43
           "steps_to_run": ["Align+Transcribe", "Raw Synthesis", "Measure & Build SSML", "Synthesize+Merge", "Export JSON", "Final Transcribe"]}
    E.set_default_engine(engine)
    try:
        failed = AP.run_all(cfg, base=tmp_path)
    finally:
        E.set_default_engine(None)
    assert failed == []
    voice = tmp_path / "Data" / "voice" / VOICE
    res_dir = tmp_path / "Out" / "results" / VOICE
    # ---- step 1 left the reference's folders for all ten recordings
    segs = []
    n_words = 0
    for n in names:
        tg = TG.read_textgrid(voice / "WhisperTS_textgrid_files" / f"{n}.TextGrid")
        assert len(tg.tiers) == 1 and tg.tiers[0].name == "words" and tg.tiers[0].intervals
        assert (voice / "transcription" / f"{n}.txt").exists() and (voice / "transcription_raw" / f"{n}.txt").exists()
        dur = len(pcm[("nat", n)]) / nat_rate
        n_iv = len(tg.tiers[0].intervals)
        assert tg.tiers[0].intervals[-1][1] <= dur + 0.05 + 0.02 * n_iv           # (a random model parks words on the last frame: 20 ms each behind it)
        segs.append(T.SegmentInput(n, tg.tiers[0].intervals))
        n_words += sum(1 for _, _, m in tg.tiers[0].intervals if m.strip())
    long_tg = TG.read_textgrid(voice / "WhisperTS_textgrid_files" / "segment_ph6.TextGrid")
    if [m for _, _, m in long_tg.tiers[0].intervals] != ["..."]:
        assert long_tg.tiers[0].intervals[-1][1] > 30.0                        # the 37.2 s recording went through its second window
    assert n_words >= 30, {n: [m for _, _, m in TG.read_textgrid(voice / "WhisperTS_textgrid_files" / f"{n}.TextGrid").tiers[0].intervals][:12] for n in names}
    # ---- step 2: the three tables against the CPU path on the same TextGrids
    ap = AP.AudioPipeline(VOICE, cfg, base=tmp_path, engine=engine)
    want = T.SsmlTagger(ap.settings, ap.azure_voice).run(segs, MixedRateOracle(pcm, rates, nat_rate))
    got = ap.measure_prosody_and_build_ssml()                                  # (the same step again, for the numbers behind the strings)
    assert len(want.rows) >= 20
    for a, b in zip(got.segment_stats, want.segment_stats):
        assert a["segment"] == b["segment"] and a["wc"] == b["wc"] and a["d_nat"] == b["d_nat"] and a["d_syn"] == b["d_syn"]
        assert abs(a["p_nat"] - b["p_nat"]) <= 1e-6 * max(b["p_nat"], 1.0)
        assert abs(a["l_nat"] - b["l_nat"]) <= 1e-6 and abs(a["l_syn"] - b["l_syn"]) <= 1e-6
    for a, b in zip(got.rows, want.rows):
        assert (a["segment"], a["syntagme"], a["pause"]) == (b["segment"], b["syntagme"], b["pause"])
    for name, df in (("BDD_ssml.csv", want.bdd_ssml), ("BDD_syntagme_ssml.csv", want.bdd_syntagme_ssml), ("BDD_syntagme_for_synth.csv", want.bdd_syntagme_for_synth)):
        p = tmp_path / ("want_" + name)
        df.to_csv(p, index=False)
        assert (res_dir / name).read_text(encoding="utf-8") == p.read_text(encoding="utf-8"), name
    # ---- the path the ranks of a multi-GPU run take (local block, run_sharded, the records through allgather_records) at world size 1: same three files
    sharded = AP.AudioPipeline(VOICE, dict(cfg, out_dir="Out_sharded", force_sharded_path=True), base=tmp_path, engine=engine)
    sharded.measure_prosody_and_build_ssml()
    for name in ("BDD_ssml.csv", "BDD_syntagme_ssml.csv", "BDD_syntagme_for_synth.csv"):
        assert (sharded.results_dir / name).read_text(encoding="utf-8") == (res_dir / name).read_text(encoding="utf-8"), name
    # ---- the last step: OUT.wav transcribed, results moved beside it; the run's configuration recorded
    assert (res_dir / "OUT.TextGrid").exists() and (res_dir / "OUT.txt").exists() and (res_dir / "used_config.yaml").exists()
    # ---- break prediction over the voice's cleaned transcriptions
    dims = dict(BW.DIMS["tiny"], n_pos=128)                                   # (pause_bert.py:16 truncates at 128 tokens)
    piece = lambda w: [3 + (sum(map(ord, w)) % 250), 5 + len(w) % 200][: 1 + len(w) % 2]
    labels = ap.predict_breaks(word_piecer=piece, weights=BW.synthetic_weights(dims, seed=5), dims=dims, cls_id=1, sep_id=2)
    assert list(labels) == names
    for n in names:
        assert len(labels[n]) == len((voice / "transcription" / f"{n}.txt").read_text(encoding="utf-8").split())
    assert (res_dir / "BDD_breaks.csv").read_text(encoding="utf-8").splitlines()[0] == "segment,word_index,word,break"
