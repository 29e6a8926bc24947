"""tiny-BERT for masked-LM (BASELINE config #5): model classes with the parameter naming of the public
BERT checkpoints (so `load_parameters` takes a HuggingFace-style state dict), written against the tensor
API only - they run on CpuTensor and on HipTensor.  Counterpart of the model half of the reference's
examples/bert.py:14-229; its tokenizer and `from_pretrained` need network access and are out of scope.

Differences from the reference, on purpose:
  * embeddings are looked up with `weight[ids]` on the tensor's own backend (the reference round-trips
    through the CPU and thereby drops the embedding gradient, bert.py:19-21);
  * `gelu` uses the backend's fused op when there is one (same expression, bert.py:12); so does the scaling of the
    attention scores in front of their softmax (bert.py:81-86) when there is no mask to add in between, and the two residual
    additions of a layer (bert.py:101, :117), spelled `dense(h, residual=r)` (nn.Linear adds r where the product is made).

    python examples/bert.py [--cpu] [--batch 8]        # forward + backward of a random tiny-BERT
"""
import math
import os
import sys
import time
import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import lightgrad_amd as light  # noqa: E402
import lightgrad_amd.nn as nn  # noqa: E402

TINY = dict(hidden_size=128, intermediate_size=512, num_hidden_layers=2, num_attention_heads=2,
            vocab_size=30522, max_position_embeddings=512, type_vocab_size=2)


def gelu(x):
    if hasattr(x, "gelu"):
        return x.gelu()
    return 0.5 * x * (1.0 + (x * 0.7978845608 * (1.0 + 0.044715 * x * x)).tanh())


class Embedding(nn.Module):
    def __init__(self, embedding_dim, vocab_size):
        nn.Module.__init__(self)
        self.weight = light.xavier((vocab_size, embedding_dim))

    def forward(self, ids):
        return self.weight[ids]


class BertEmbedding(nn.Module):
    def __init__(self, hidden_size, vocab_size, max_position_embeddings, type_vocab_size):
        nn.Module.__init__(self)
        self.word_embeddings = Embedding(hidden_size, vocab_size)
        self.position_embeddings = Embedding(hidden_size, max_position_embeddings)
        self.token_type_embeddings = Embedding(hidden_size, type_vocab_size)
        self.LayerNorm = nn.LayerNorm(hidden_size)
        self._const_ids = {}        # (backend class, shape) -> constant id tensors, uploaded once

    def _constant_ids(self, cls, shape):
        key = (cls, tuple(shape))
        if key not in self._const_ids:
            self._const_ids[key] = (cls.from_numpy(np.zeros(shape, dtype=np.int32), requires_grad=False),
                                    cls.from_numpy(np.arange(shape[-1], dtype=np.int32), requires_grad=False))
            for t in self._const_ids[key]:
                if hasattr(t, "freeze"):
                    t.freeze()              # constants of the model: a backend may keep what it derives from them (and record it in graphs)
        return self._const_ids[key]

    def forward(self, input_ids, token_type_ids=None):
        zeros, position_ids = self._constant_ids(input_ids.__class__, input_ids.shape)
        if token_type_ids is None:
            token_type_ids = zeros
        word, position, kind = self.word_embeddings.weight, self.position_embeddings.weight, self.token_type_embeddings.weight
        if hasattr(word, "embedding_sum"):
            # the backend's one-kernel form of the line below: three lookups and their sum
            e = word.embedding_sum(position, kind, ids0=input_ids, ids1=position_ids, ids2=token_type_ids)
        else:
            e = self.word_embeddings(input_ids) + self.position_embeddings(position_ids) + self.token_type_embeddings(token_type_ids)
        return self.LayerNorm(e)


class BertSelfAttention(nn.Module):
    def __init__(self, hidden_size, num_attention_heads):
        nn.Module.__init__(self)
        assert hidden_size % num_attention_heads == 0
        self.h, self.d = num_attention_heads, hidden_size // num_attention_heads
        self.query = nn.Linear(hidden_size, hidden_size)
        self.key = nn.Linear(hidden_size, hidden_size)
        self.value = nn.Linear(hidden_size, hidden_size)

    def forward(self, hidden, attention_mask=None):
        b, s, _ = hidden.shape
        if attention_mask is None and hasattr(hidden, "self_attention") and hidden.self_attention_supported(self.query.weight, self.h):
            # the backend's one-node form of this whole method: one launch for the three projections, one for the attention
            context = hidden.self_attention(self.query.weight, self.query.bias, self.key.weight, self.key.bias,
                                            self.value.weight, self.value.bias, heads=self.h, scale=math.sqrt(self.d) ** -1)
            return context, context.attention_probs
        q, k, v = self.query(hidden), self.key(hidden), self.value(hidden)
        if attention_mask is None and hasattr(q, "attention") and q.attention_supported(self.h):
            # the backend's one-launch form of everything below (scores, scaling, softmax, context), forward and backward
            context = q.attention(k, v, heads=self.h, scale=math.sqrt(self.d) ** -1)
            return context, context.attention_probs
        # head split: (b, s, h*d) -> (b, h, s, d) as stride permutations, no copies
        q = q.reshape(b, s, self.h, self.d).transpose(0, 2, 1, 3)
        k = k.reshape(b, s, self.h, self.d).transpose(0, 2, 3, 1)
        v = v.reshape(b, s, self.h, self.d).transpose(0, 2, 1, 3)
        scores = q @ k
        if attention_mask is None and hasattr(scores, "scaled_softmax"):
            # the backend's one-kernel form of the two lines below: the same x * sqrt(d)**-1, rounded to fp32, then softmax
            probs = scores.scaled_softmax(math.sqrt(self.d) ** -1)
        else:
            scores = scores / math.sqrt(self.d)
            if attention_mask is not None:
                # (b, s) -> (b, 1, 1, s); for batch 1 this is the reference's (1, 1, 1, s) (bert.py:82, which breaks for b > 1)
                mask = attention_mask.reshape(attention_mask.shape[0], 1, 1, attention_mask.shape[1])
                scores = scores + ((1.0 - mask) * -10000.0).detach()
            probs = scores.softmax(axis=-1)
        context = (probs @ v).transpose(0, 2, 1, 3).reshape(b, s, self.h * self.d)
        return context, probs


class BertAttention(nn.Module):
    def __init__(self, hidden_size, num_attention_heads):
        nn.Module.__init__(self)
        self.self = BertSelfAttention(hidden_size, num_attention_heads)
        self.output = nn.Module()
        self.output.dense = nn.Linear(hidden_size, hidden_size)
        self.output.LayerNorm = nn.LayerNorm(hidden_size)

    def forward(self, hidden_in, attention_mask=None):
        hidden, probs = self.self(hidden_in, attention_mask=attention_mask)
        hidden = self.output.LayerNorm(self.output.dense(hidden, residual=hidden_in))     # dense(hidden) + hidden_in
        return hidden, probs


class BertLayer(nn.Module):
    def __init__(self, hidden_size, intermediate_size, num_attention_heads):
        nn.Module.__init__(self)
        self.attention = BertAttention(hidden_size, num_attention_heads)
        self.intermediate = nn.Module()
        self.intermediate.dense = nn.Linear(hidden_size, intermediate_size)
        self.output = nn.Module()
        self.output.dense = nn.Linear(intermediate_size, hidden_size)
        self.output.LayerNorm = nn.LayerNorm(hidden_size)

    def forward(self, hidden, attention_mask=None):
        hidden, probs = self.attention(hidden, attention_mask)
        if hasattr(hidden, "feed_forward"):
            # the backend's one-node form of the line below: gelu and residual in the products' epilogues, four launches for six
            up, down = self.intermediate.dense, self.output.dense
            hidden = hidden.feed_forward(up.weight, up.bias, down.weight, down.bias, hidden)
        else:
            hidden = self.output.dense(gelu(self.intermediate.dense(hidden)), residual=hidden)  # hidden + dense(...)
        return self.output.LayerNorm(hidden), probs


class BertModel(nn.Module):
    def __init__(self, hidden_size, intermediate_size, num_hidden_layers, num_attention_heads, vocab_size,
                 max_position_embeddings, type_vocab_size, **unused):
        nn.Module.__init__(self)
        self.embeddings = BertEmbedding(hidden_size, vocab_size, max_position_embeddings, type_vocab_size)
        self.encoder = nn.Module()
        self.encoder.layer = nn.ModuleList(*[BertLayer(hidden_size, intermediate_size, num_attention_heads)
                                             for _ in range(num_hidden_layers)])

    def forward(self, input_ids, attention_mask=None, token_type_ids=None):
        hidden = self.embeddings(input_ids, token_type_ids=token_type_ids)
        for layer in self.encoder.layer:
            hidden, _ = layer(hidden, attention_mask=attention_mask)
        return hidden


class BertForMaskedLM(nn.Module):
    def __init__(self, hidden_size, vocab_size, **config):
        nn.Module.__init__(self)
        self.bert = BertModel(hidden_size=hidden_size, vocab_size=vocab_size, **config)
        self.cls = nn.Module()
        self.cls.predictions = nn.Module()
        self.cls.predictions.transform = nn.Module()
        self.cls.predictions.transform.dense = nn.Linear(hidden_size, hidden_size)
        self.cls.predictions.transform.LayerNorm = nn.LayerNorm(hidden_size)
        self.cls.predictions.decoder = nn.Linear(hidden_size, vocab_size, bias=False)
        self.cls.predictions.bias = light.zeros(vocab_size)

    def forward(self, input_ids, attention_mask=None, token_type_ids=None):
        h = self.bert(input_ids=input_ids, attention_mask=attention_mask, token_type_ids=token_type_ids)
        t = self.cls.predictions.transform
        h = t.LayerNorm(gelu(t.dense(h)))
        return self.cls.predictions.decoder(h) + self.cls.predictions.bias


def forward_backward(model, ids):
    logits = model(ids)
    loss = (logits * logits).mean()
    for p in model.parameters():
        p.zero_grad()
    loss.backward()
    return loss


if __name__ == "__main__":
    cpu = "--cpu" in sys.argv
    batch = int(sys.argv[sys.argv.index("--batch") + 1]) if "--batch" in sys.argv else 8
    to_device = (lambda t: t) if cpu else (lambda t: t.hip())
    np.random.seed(0)
    model = BertForMaskedLM(**TINY).map_parameters(to_device)
    ids = to_device(light.from_numpy(np.random.randint(0, TINY["vocab_size"], (batch, 128)).astype(np.int32), requires_grad=False))
    for it in range(3):
        t0 = time.perf_counter()
        value = forward_backward(model, ids).item()            # .item() synchronises
        print("iter %d: loss %.6f  fwd+bwd %.1f ms (eager tape)" % (it, value, 1e3 * (time.perf_counter() - t0)))
    if not cpu and "--graph" in sys.argv:
        # static shapes: record the ~400 launches of one forward+backward once, replay them with one host call
        from lightgrad_amd.autograd.hip import HipGraph, HipDevice
        graph = HipGraph()
        with graph.capture():
            loss = forward_backward(model, ids)
        replays = int(sys.argv[sys.argv.index("--replays") + 1]) if "--replays" in sys.argv else 3
        for it in range(replays):
            HipDevice.synchronize()
            t0 = time.perf_counter()
            graph.replay()
            value = loss.item()
            if it < 3 or it == replays - 1:
                print("replay %d: loss %.6f  fwd+bwd %.2f ms (hipGraph)" % (it, value, 1e3 * (time.perf_counter() - t0)))
