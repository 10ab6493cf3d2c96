"""GPU parity of the break-prediction token classifier (BertForTokenClassification forward).
Tolerances: bf16 MFMA operands with fp32 accumulation / LayerNorm / residual stream: logits within 0.08 absolute
(observed 0.02 on logits of magnitude 2) and 3e-2 relative L2 of the transformers / torch fp32 values; labels equal
wherever the fp32 margin between the two classes exceeds 0.15."""
import numpy as np
import pytest

from oracle import bert_oracle as BO
from prosody_control_french_tts_amd import PceError, ProsodyEngine, bert_weights as BW
from tests.test_bert import load_gold

pytestmark = pytest.mark.gpu


def _check(got, want):
    logits, labels = got
    assert logits.shape == want.shape and logits.dtype == np.float32
    assert np.max(np.abs(logits - want)) <= 0.08, float(np.max(np.abs(logits - want)))
    assert np.linalg.norm(logits - want) <= 3e-2 * max(np.linalg.norm(want), 1.0)
    assert np.array_equal(labels, np.argmax(logits, axis=1))
    sure = np.abs(want[:, 1] - want[:, 0]) > 0.15
    assert np.array_equal(labels[sure], np.argmax(want, axis=1)[sure])


def test_tiny_model_matches_transformers_logits(engine):
    dims, W, toks, logits, lens = load_gold()
    engine.bert_load(dims, BW.pack(W, dims))
    res = engine.bert_token_classify(toks)
    for i, n in enumerate(lens):
        _check(res[i], logits[i, :n])
    # batch composition does not matter: one sequence alone gives the same bits as inside the batch
    alone = engine.bert_token_classify([toks[3]])[0]
    assert alone[0].tobytes() == res[3][0].tobytes()


def test_base_architecture_matches_torch_restatement(engine):
    dims = dict(BW.DIMS["mbert-base-uncased"], n_vocab=2000)            # the real depth / width / heads; a smaller vocabulary table
    W = BW.synthetic_weights(dims, seed=11)
    rng = np.random.default_rng(2)
    lens = [128, 1, 37, 64, 65, 100, 128, 5] + rng.integers(2, 129, size=24).tolist()
    toks = [rng.integers(0, dims["n_vocab"], size=n).tolist() for n in lens]
    engine.bert_load(dims, BW.pack(W, dims))
    res = engine.bert_token_classify(toks)
    want = BO.forward(toks, W, dims)
    for g, w in zip(res, want):
        _check(g, w)


def test_bert_argument_errors():
    with ProsodyEngine(0) as eng:
        with pytest.raises(PceError):
            eng.bert_run([[1, 2, 3]])                                   # before load
        dims = BW.DIMS["tiny"]
        eng.bert_load(dims, BW.pack(BW.synthetic_weights(dims), dims))
        with pytest.raises(PceError):
            eng.bert_run([[1, 2, dims["n_vocab"]]])                     # id out of the vocabulary
        with pytest.raises(PceError):
            eng.bert_run([list(range(dims["n_pos"] + 1))])              # longer than the position table
        with pytest.raises(PceError):
            eng.bert_run([[]])
        with pytest.raises(PceError):
            eng.bert_load(dims, np.zeros(10, np.float32))               # wrong blob size
        eng.bert_run([])


def test_predict_breaks_reads_the_first_subtoken(engine):
    from prosody_control_french_tts_amd.Preprocessing import break_bert as BB
    dims, W, _, _, _ = load_gold()
    engine.bert_load(dims, BW.pack(W, dims))
    rng = np.random.default_rng(4)
    sents = [[rng.integers(3, dims["n_vocab"], size=int(rng.integers(1, 4))).tolist() for _ in range(nw)] for nw in (6, 1, 30)]
    got = BB.predict_breaks(engine, sents, cls_id=1, sep_id=2, max_length=dims["n_pos"])
    for s, g in zip(sents, got):
        ids, wids = BB.encode_words(s, 1, 2, dims["n_pos"])
        want = BO.forward([ids], W, dims)[0]
        first = BB.first_subtoken_positions(wids, len(s))
        assert len(g) == len(s)
        for w, p in enumerate(first):
            if p is None:
                assert g[w] == 0
            elif abs(want[p, 1] - want[p, 0]) > 0.15:
                assert g[w] == int(np.argmax(want[p]))


def test_audio_pipeline_predict_breaks_on_the_engine(engine, tmp_path):
    """BASELINE.json configs[4]'s break-prediction forward reached from ``AudioPipeline`` (world size 1 here; the sharding over ranks
    and its one all-gather run under gloo in tests/test_abi_and_shard.py): one sentence per segment from the voice's cleaned
    transcriptions, labels = the first sub-token's arg-max per word, the table written as BDD_breaks.csv."""
    from prosody_control_french_tts_amd.audio_pipeline import AudioPipeline
    from prosody_control_french_tts_amd.Preprocessing import break_bert as BB
    dims = BW.DIMS["tiny"]
    W = BW.synthetic_weights(dims, seed=5)
    cfg = {"data_dir": "Data", "out_dir": "Out", "whisper_device": "cuda:0", "steps_to_run": []}
    ap = AudioPipeline("v1", cfg, base=tmp_path, engine=engine)
    ap.transcription_dir.mkdir(parents=True)
    texts = {"segment_ph1": "bonjour tout le monde", "segment_ph2": "oui", "segment_ph10": "une phrase nettement plus longue que les autres ici"}
    for k, t in texts.items():
        (ap.transcription_dir / f"{k}.txt").write_text(t, encoding="utf-8")
    piece = lambda w: [3 + (sum(map(ord, w)) % 250), 5 + len(w)][: 1 + len(w) % 2]
    got = ap.predict_breaks(word_piecer=piece, weights=W, dims=dims, cls_id=1, sep_id=2)
    assert list(got) == ["segment_ph1", "segment_ph2", "segment_ph10"]                      # segment order, as every table of the pipeline
    want = BB.predict_breaks(engine, [[piece(w) for w in texts[k].split()] for k in got], 1, 2)
    assert [got[k] for k in got] == want and all(len(got[k]) == len(texts[k].split()) for k in got)
    rows = (ap.results_dir / "BDD_breaks.csv").read_text(encoding="utf-8").splitlines()
    assert rows[0] == "segment,word_index,word,break" and len(rows) == 1 + sum(len(t.split()) for t in texts.values())
