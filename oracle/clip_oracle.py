"""CPU restatement of the CLIP text transformer behind the reference's ``FrozenCLIPEmbedder`` (libs/clip.py:40-91:
``CLIPTextModel(input_ids=tokens).last_hidden_state``; the model itself is Hugging Face ``transformers``'
``CLIPTextTransformer``: token + position embedding, pre-LN blocks with causal multi-head attention and a
quick-GELU MLP, final LayerNorm).  TEST INFRASTRUCTURE ONLY -- never imported by the product path.
Pinned against ``tests/golden/clip_text_tiny.npz`` (the HF module itself, tiny configuration, random init)."""
import numpy as np

from . import _cops as C


def _get(sd, key):
    return sd[key] if key in sd else sd["text_model." + key]


def causal_attention(q, k, v, heads):
    """q, k, v: [B, L, D]; softmax(q k^T / sqrt(dh) + causal mask) v per head, fp32."""
    B, L, D = q.shape
    dh = D // heads
    qh = q.reshape(B, L, heads, dh).transpose(0, 2, 1, 3)
    kh = k.reshape(B, L, heads, dh).transpose(0, 2, 1, 3)
    vh = v.reshape(B, L, heads, dh).transpose(0, 2, 1, 3)
    s = np.einsum("bhqd,bhkd->bhqk", qh, kh).astype(np.float32) * np.float32(dh ** -0.5)
    s = np.where(np.tril(np.ones((L, L), bool))[None, None], s, -np.inf)
    s = s - s.max(-1, keepdims=True)
    p = np.exp(s)
    p = p / p.sum(-1, keepdims=True)
    o = np.einsum("bhqk,bhkd->bhqd", p, vh)
    return o.transpose(0, 2, 1, 3).reshape(B, L, D).astype(np.float32)


def quick_gelu(x):
    return (x / (1.0 + np.exp(-1.702 * x))).astype(np.float32)


def text_forward(sd, ids, heads, eps=1e-5, hidden=None):
    """ids [B, L] int -> last_hidden_state [B, L, D]; ``hidden`` (list) receives the state after the embeddings and
    after every layer, like HF's ``output_hidden_states``."""
    ids = np.asarray(ids)
    B, L = ids.shape
    x = (_get(sd, "embeddings.token_embedding.weight")[ids] +
         _get(sd, "embeddings.position_embedding.weight")[None, :L]).astype(np.float32)
    if hidden is not None:
        hidden.append(x.copy())
    i = 0
    while f"encoder.layers.{i}.layer_norm1.weight" in sd or f"text_model.encoder.layers.{i}.layer_norm1.weight" in sd:
        pre = f"encoder.layers.{i}."
        h = C.layernorm(x, _get(sd, pre + "layer_norm1.weight"), _get(sd, pre + "layer_norm1.bias"), eps=eps)
        q = C.linear(h, _get(sd, pre + "self_attn.q_proj.weight"), _get(sd, pre + "self_attn.q_proj.bias"))
        k = C.linear(h, _get(sd, pre + "self_attn.k_proj.weight"), _get(sd, pre + "self_attn.k_proj.bias"))
        v = C.linear(h, _get(sd, pre + "self_attn.v_proj.weight"), _get(sd, pre + "self_attn.v_proj.bias"))
        a = causal_attention(q, k, v, heads)
        x = x + C.linear(a, _get(sd, pre + "self_attn.out_proj.weight"), _get(sd, pre + "self_attn.out_proj.bias"))
        h = C.layernorm(x, _get(sd, pre + "layer_norm2.weight"), _get(sd, pre + "layer_norm2.bias"), eps=eps)
        h = quick_gelu(C.linear(h, _get(sd, pre + "mlp.fc1.weight"), _get(sd, pre + "mlp.fc1.bias")))
        x = x + C.linear(h, _get(sd, pre + "mlp.fc2.weight"), _get(sd, pre + "mlp.fc2.bias"))
        if hidden is not None:
            hidden.append(x.copy())
        i += 1
    return C.layernorm(x, _get(sd, "final_layer_norm.weight"), _get(sd, "final_layer_norm.bias"), eps=eps)
