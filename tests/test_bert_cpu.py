"""tiny-BERT (BASELINE config #5) on the CPU backend against the forward logits recorded from the reference
(tests/golden/bert_tiny_forward.npz).  The reference has no working BERT backward: gradients
are pinned by numerical differentiation on a reduced model."""
import importlib.util
import os
import numpy as np
import pytest
import lightgrad_amd as light
from lightgrad_amd import CpuTensor
from lightgrad_amd.autograd.utils.gradcheck import assert_gradcheck
from conftest import ROOT, load_golden

spec = importlib.util.spec_from_file_location("bert_example", os.path.join(ROOT, "examples", "bert.py"))
bert = importlib.util.module_from_spec(spec)
spec.loader.exec_module(bert)


def build_tiny(seed=42):
    np.random.seed(seed)
    return bert.BertForMaskedLM(**bert.TINY)


def test_forward_matches_reference_fixture():
    g = load_golden("bert_tiny_forward.npz")
    model = build_tiny()
    names = sorted(n for n, _ in model.named_parameters())
    assert names == list(g["param_names"])                      # same parameter naming as the reference
    sums = {n: float(np.abs(p.numpy().astype(np.float64)).sum()) for n, p in model.named_parameters()}
    np.testing.assert_allclose([sums[n] for n in names], g["param_abs_sums"], rtol=1e-12)    # same init stream
    with light.no_grad():
        logits = model(CpuTensor.from_numpy(g["ids"], requires_grad=False)).numpy()
        masked = model(CpuTensor.from_numpy(g["ids"][:1], requires_grad=False),
                       attention_mask=CpuTensor.from_numpy(g["mask"], requires_grad=False)).numpy()
    assert logits.shape == (2, 128, 30522)
    np.testing.assert_allclose(logits[:, :, ::509], g["logits_sample"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(masked[:, ::8, ::509], g["logits_masked_sample"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose([logits.astype(np.float64).sum(), np.abs(logits).astype(np.float64).sum()], g["logits_digest"], rtol=1e-6)
    assert (logits.argmax(-1) == g["argmax"]).mean() > 0.999


def small_model(seed=3):
    np.random.seed(seed)
    return bert.BertForMaskedLM(hidden_size=8, intermediate_size=16, num_hidden_layers=1, num_attention_heads=2,
                                vocab_size=11, max_position_embeddings=6, type_vocab_size=2)


def test_backward_by_numerical_differentiation():
    """gradient of the logits w.r.t. the word-embedding table (through residuals, LayerNorm, softmax, gelu,
    batched attention GEMMs and a REPEATED token id) and w.r.t. one attention weight"""
    model = small_model()
    ids = CpuTensor.from_numpy(np.array([[1, 4, 4, 7]], dtype=np.int32), requires_grad=False)
    emb = model.bert.embeddings.word_embeddings

    def f_table(w):
        emb.weight = w
        return model(ids)[0, :, ::3]
    assert_gradcheck(f_table, CpuTensor.from_numpy(emb.weight.numpy().astype(np.float32)), eps=1e-2, atol=3e-3, rtol=3e-2)
    q = model.bert.encoder.layer[0].attention.self.query

    def f_query(w):
        q.weight = w
        return model(ids)[0, :, ::3]
    assert_gradcheck(f_query, CpuTensor.from_numpy(q.weight.numpy().astype(np.float32)), eps=1e-2, atol=3e-3, rtol=3e-2)


def test_embedding_gradient_accumulates_repeated_ids():
    w = CpuTensor.from_numpy(np.arange(12, dtype=np.float32).reshape(4, 3))
    ids = CpuTensor.from_numpy(np.array([2, 0, 2, 2], dtype=np.int64), requires_grad=False)
    w[ids].backward(allow_fill=True)
    np.testing.assert_array_equal(w.grad.numpy(), [[1, 1, 1], [0, 0, 0], [3, 3, 3], [0, 0, 0]])
