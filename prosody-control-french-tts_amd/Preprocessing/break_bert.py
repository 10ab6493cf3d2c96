"""Inference path of the break-prediction token classifier the reference trains
(Code/baseline_models/pause_bert.py: ``BertForTokenClassification`` on bert-base-multilingual-uncased, label 1 = a
break follows the word, only the FIRST sub-token of a word carries the label, ``:76-91``; the reference stops at
training and evaluation, this is the forward pass a pipeline step would call).

Tokenisation proper needs the checkpoint's WordPiece vocabulary, which is not reachable offline: the functions here
take every word already split into sub-token ids (``tokenizer(words, is_split_into_words=True)`` does exactly that
split) and reproduce the rest of the encoding: [CLS] ... [SEP], truncation to ``max_length`` (``:66-73``), the
word_ids bookkeeping and the first-sub-token read-out.  The array work is ``ProsodyEngine.bert_*`` (libpce.so)."""
from __future__ import annotations

from ..bert_weights import MAX_LENGTH


def encode_words(word_pieces, cls_id: int, sep_id: int, max_length: int = MAX_LENGTH):
    """-> (input_ids, word_ids): word_ids[i] is the index of the word token i belongs to, None for [CLS] / [SEP]."""
    ids, wids = [cls_id], [None]
    budget = max_length - 2
    for wi, pieces in enumerate(word_pieces):
        for p in pieces:
            if len(ids) - 1 >= budget:
                break
            ids.append(int(p)); wids.append(wi)
    ids.append(sep_id); wids.append(None)
    return ids, wids


def first_subtoken_positions(word_ids, n_words: int):
    """Position of the first sub-token of every word (None when truncation removed the word)."""
    first = [None] * n_words
    prev = None
    for i, w in enumerate(word_ids):
        if w is not None and w != prev and first[w] is None:
            first[w] = i
        prev = w
    return first


def predict_breaks(engine, sentences, cls_id: int, sep_id: int, max_length: int = MAX_LENGTH):
    """``sentences``: list of sentences, each a list of words, each a list of sub-token ids.
    -> per sentence a list of 0 / 1 per word (1 = break after the word; 0 for words cut off by truncation)."""
    enc = [encode_words(s, cls_id, sep_id, max_length) for s in sentences]
    engine.bert_run([e[0] for e in enc])
    out = []
    for i, (s, (_, wids)) in enumerate(zip(sentences, enc)):
        _, labels = engine.bert_fetch(i)
        first = first_subtoken_positions(wids, len(s))
        out.append([int(labels[p]) if p is not None else 0 for p in first])
    return out
