"""``Aligners/transcribe.py``'s restatement of ``whisper.transcribe``'s window loop (seek from the timestamp tokens, prompts with
``condition_on_previous_text``, segment times) against transformers' long-form ``generate`` (golden
tests/golden/transcribe_hf_long.json, made by tests/golden/make_goldens_transcribe_hf.py with the installed transformers): the same
random-init two-layer model, recordings of 39 s and 66 s, segment by segment the same token ids and the same times.

CPU test: the engine is replaced by a stub that answers every call ``transcribe_batch`` makes from the float32 restatement
(oracle/whisper_oracle.py) -- the loop under test is host logic; the kernels behind the same calls are checked against the same
restatement in the GPU tests.  Window features: a slice of the whole recording's log-mel, zero-filled past the content, which is how
transformers cuts its windows (openai-whisper pads the AUDIO with 30 s of zeros instead; that front end has its own tests)."""
import json
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
from oracle import whisper_oracle as WO  # noqa: E402
from prosody_control_french_tts_amd.Aligners import checkpoint as CK, transcribe as TR  # noqa: E402


class OracleEngine:
    """The calls of ``transcribe_batch`` (vad=None) answered on the CPU by the float32 restatement."""

    def __init__(self, Wd, tdims, We, edims, tok):
        self.Wd, self.tdims, self.We, self.edims, self.tok = Wd, tdims, We, edims, tok
        self.enc = []

    def upload(self, clips, rate):
        import make_goldens_transcribe_hf as G
        self.clips = [np.asarray(c) for c in clips]
        self.clip_lengths = [len(c) for c in self.clips]
        self.mels = [G.mel_full(c) for c in self.clips]

    def logmel_run_at(self, n_mels, seeks):
        self.windows = []
        for m, s in zip(self.mels, seeks):
            w = np.zeros((n_mels, 3000), np.float32)
            seg = m[:, int(s):int(s) + 3000]
            w[:, :seg.shape[1]] = seg
            self.windows.append(w)

    def whisper_encode_run(self):
        self.enc = [WO.encoder_forward(w, self.We, self.edims) for w in self.windows]

    def whisper_sample_keys(self, keys=None):
        self.sample_keys = None if keys is None else list(keys)

    def whisper_num_encoded(self):
        return len(self.enc)

    def _rules(self, eot, ts_begin, mask, max_init):
        m = np.asarray(mask)
        return dict(eot=eot, timestamp_begin=ts_begin, no_timestamps=self.tok.no_timestamps, max_initial_timestamp_index=max_init,
                    suppress_tokens=[int(i) for i in np.nonzero(m & 1)[0] if i != self.tok.no_timestamps], blank_tokens=[int(i) for i in np.nonzero(m & 2)[0]])

    def whisper_decode_step_ex(self, token_lists, sample_begin, eot, timestamp_begin, vocab_mask, max_initial_timestamp_index=None, temperature=0.0,
                               seed=0, probe_token=-1, no_cache=False):
        assert temperature == 0.0
        rules = self._rules(eot, timestamp_begin, vocab_mask, max_initial_timestamp_index)
        n = len(token_lists)
        sb = [int(sample_begin)] * n if np.isscalar(sample_begin) else [int(x) for x in sample_begin]
        nxt, lp, pr = np.zeros(n, np.int32), np.zeros(n, np.float32), (np.zeros(n, np.float32) if probe_token >= 0 else None)
        for i, p in enumerate(token_lists):
            p = [int(t) for t in p]
            logits = WO.find_alignment(p, self.enc[i], self.Wd, self.tdims, 2, 0, want_internal=True)["logits"][-1].astype(np.float64)
            if pr is not None:
                e = np.exp(logits - logits.max()); pr[i] = e[probe_token] / e.sum()
            if len(p) > sb[i] and p[-1] == eot:
                nxt[i] = eot; continue
            f = WO.apply_decoding_rules(logits.astype(np.float32), p, sb[i], rules).astype(np.float64)
            k = int(np.argmax(f)); fin = np.isfinite(f)
            nxt[i] = k; lp[i] = f[k] - (f[fin].max() + np.log(np.sum(np.exp(f[fin] - f[fin].max()))))
        return nxt, lp, pr

    def whisper_decode_loop(self, token_lists, sample_begin, eot, timestamp_begin, vocab_mask, max_new, max_initial_timestamp_index=None, temperature=0.0,
                            seed=0, probe_token=-1, no_cache=False, check_every=4):
        seqs = [list(map(int, p)) for p in token_lists]
        toks, lps = [], []
        for _ in range(int(max_new)):
            nxt, lp, _ = self.whisper_decode_step_ex(seqs, sample_begin, eot, timestamp_begin, vocab_mask, max_initial_timestamp_index)
            toks.append(nxt); lps.append(lp)
            for s_, t in zip(seqs, nxt):
                s_.append(int(t))
            if all(s_[-1] == eot for s_ in seqs):
                break
        return np.stack(toks, 1), np.stack(lps, 1), None

    def whisper_align(self, token_lists, num_frames, sot_len, head_mask=None, **kw):
        out = []
        for i, (t, nf) in enumerate(zip(token_lists, num_frames)):
            _, ti, tj = WO.find_alignment([int(x) for x in t], self.enc[i], self.Wd, self.tdims, int(nf), int(sot_len))
            out.append({"text_indices": ti, "time_indices": tj})
        return out


@pytest.fixture(scope="module")
def setup():
    import make_goldens_transcribe_hf as G
    gold = json.load(open(os.path.join(HERE, "golden", "transcribe_hf_long.json")))
    We, Wd = G.weights()
    model = CK.WhisperModel({k: v for k, v in We.items()}, {k: v for k, v in Wd.items()}, alignment_heads=[[1, 0], [1, 1]], name="toy")
    return G, gold, We, Wd, model


@pytest.mark.parametrize("case", range(7))
def test_window_loop_matches_transformers_long_form_generate(setup, case):
    G, gold, We, Wd, model = setup
    c = gold["cases"][case]
    eng = OracleEngine(Wd, G.TDIMS, We, G.EDIMS, G.TOK)
    opts = TR.TranscribeOptions(vad=None, condition_on_previous_text=c["condition_on_previous_text"], temperature=(0.0,), sample_len=gold["sample_len"],
                                logprob_threshold=c["logprob_threshold"], no_speech_threshold=c["no_speech_threshold"], compression_ratio_threshold=None,
                                detect_disfluencies=False)
    res = TR.transcribe_batch(eng, model, G.TOK, [G.clip(c["clip"])], opts)[0]
    got = [(round(s["start"], 2), round(s["end"], 2), list(s["tokens"])) for s in res["segments"]]
    # transcribe_batch drops segments whose text is empty (whisper.transcribe does: `if not text.strip(): clear`); so does the comparison
    def canon(t):
        # transformers keeps the SECOND timestamp of a closing pair inside the window's last segment ("to know it was no single ending",
        # generation_whisper.py); whisper.transcribe's slices end on the first (tokens[last_slice:current_slice]): same segment, same times
        return t[:-1] if len(t) >= 2 and t[-1] == t[-2] and t[-1] >= G.TOK.timestamp_begin else t
    want = [(round(s["start"], 2), round(s["end"], 2), canon(s["tokens"])) for s in c["segments"]
            if G.TOK.decode([t for t in s["tokens"] if t < G.TOK.eot]).strip()]
    assert [g[2] for g in got] == [w[2] for w in want], (got, want)
    for g, w in zip(got, want):
        assert abs(g[0] - w[0]) <= 0.011 and abs(g[1] - w[1]) <= 0.011, (g, w)
    assert len(got) == len(want) and (c["no_speech_threshold"] is not None or len(got) >= 4)
