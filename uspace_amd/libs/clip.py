"""Drop-in for the reference's text conditioning encoder (libs/clip.py:40-91 ``FrozenCLIPEmbedder``): prompts ->
``[B, 77, 768]`` context for ``uvit_t2i``.  The transformer (Hugging Face ``CLIPTextModel`` in the reference) runs in
libuspace_hip.so (``uspace_clip_text_forward``); the parameters carry the HF ``state_dict`` names, so
``load_state_dict(CLIPTextModel.from_pretrained(...).state_dict())`` (with or without the ``text_model.`` prefix)
works.  Tokenisation stays on the host with the HF tokenizer, exactly as the reference does it; the module is meant to
be created once and kept (the reference re-instantiates the encoder on every call, tools/utils_t2i.py:25-39).
SURVEY.md 8(f) rank 4.
"""
import ctypes

import numpy as np
import torch
import torch.nn as nn

from .. import _hip
from ._uvit_core import ParamGroup

CLIP_L_TEXT = dict(vocab_size=49408, hidden_size=768, intermediate_size=3072, num_hidden_layers=12,
                   num_attention_heads=12, max_position_embeddings=77, layer_norm_eps=1e-5)   # openai/clip-vit-large-patch14


def _word_piece_spans(words, pieces):
    """Half-open piece ranges ``[lo, hi)`` of every word: a word takes pieces until their characters cover its length
    (at least one piece, also for an empty word); words beyond the last piece get empty ranges."""
    spans, pos = [], 0
    for w in words:
        lo, covered = pos, 0
        while pos < len(pieces):
            covered += len(pieces[pos])
            pos += 1
            if covered >= len(w):
                break
        spans.append((lo, pos))
    return spans


def get_word_inds(text, word_place, tokenizer):
    """Token positions of the selected word(s) of ``text`` in the tokenizer's output, counted with ``<bos>`` at
    position 0 -- the indices the prompt-to-prompt hooks address in the ``[B, H, L, 77]`` cross-attention maps.
    ``word_place`` is a word (every occurrence counts), a word index, or a list of word indices.
    Same results as the reference's helper of this name (libs/clip.py:6-27), pinned by
    ``tests/golden/word_inds.json``; here the words are first mapped to ranges of word pieces."""
    words = text.split(" ")
    if isinstance(word_place, str):
        wanted = [k for k, w in enumerate(words) if w == word_place]
    elif isinstance(word_place, (int, np.integer)):
        wanted = [int(word_place)]
    else:
        wanted = [int(k) for k in word_place]
    if not wanted:
        return np.array([])
    ids = tokenizer.encode(text)[1:-1]                       # without <bos> / <eos>
    pieces = [tokenizer.decode([t]).strip("#") for t in ids]
    spans = _word_piece_spans(words, pieces)
    picked = sorted({p for k in wanted if 0 <= k < len(spans) for p in range(*spans[k])})
    return np.array([p + 1 for p in picked])


def _linear(group, name, nout, nin):
    c = group.child(name)
    c.add("weight", nout, nin)
    c.add("bias", nout)


def _norm(group, name, n):
    c = group.child(name)
    c.add("weight", n)
    c.add("bias", n)


class CLIPTextTransformer(nn.Module):
    """HF CLIPTextModel's computation given token ids; parameters in the HF state_dict order and naming."""

    def __init__(self, vocab_size=49408, hidden_size=768, intermediate_size=3072, num_hidden_layers=12,
                 num_attention_heads=12, max_position_embeddings=77, layer_norm_eps=1e-5, hidden_act="quick_gelu", **_ignored):
        super().__init__()
        if hidden_act != "quick_gelu":
            raise NotImplementedError(f"hidden_act={hidden_act!r}: the CLIP text encoder the reference loads uses quick_gelu")
        if hidden_size != 64 * num_attention_heads or hidden_size % 64 or intermediate_size % 64:
            raise NotImplementedError("head_dim must be 64 and the widths multiples of 64")
        self.cfg = dict(vocab=vocab_size, dim=hidden_size, heads=num_attention_heads, layers=num_hidden_layers,
                        ffn=intermediate_size, max_pos=max_position_embeddings, eps=layer_norm_eps)
        emb = ParamGroup()
        emb.child("token_embedding").add("weight", vocab_size, hidden_size)
        emb.child("position_embedding").add("weight", max_position_embeddings, hidden_size)
        self.embeddings = emb
        enc = ParamGroup()
        layers = []
        for _ in range(num_hidden_layers):
            lyr = ParamGroup()
            att = lyr.child("self_attn")
            for n in ("k_proj", "v_proj", "q_proj", "out_proj"):
                _linear(att, n, hidden_size, hidden_size)
            _norm(lyr, "layer_norm1", hidden_size)
            mlp = lyr.child("mlp")
            _linear(mlp, "fc1", intermediate_size, hidden_size)
            _linear(mlp, "fc2", hidden_size, intermediate_size)
            _norm(lyr, "layer_norm2", hidden_size)
            layers.append(lyr)
        enc.add_module("layers", nn.ModuleList(layers))
        self.encoder = enc
        self.final_layer_norm = ParamGroup()
        self.final_layer_norm.add("weight", hidden_size)
        self.final_layer_norm.add("bias", hidden_size)
        with torch.no_grad():                                  # HF _init_weights: normal tables / projections, unit norms
            for name, prm in self.named_parameters():
                if name.endswith("norm.weight") or "layer_norm" in name and name.endswith("weight"):
                    prm.fill_(1.0)
                elif name.endswith("bias"):
                    prm.zero_()
                else:
                    prm.normal_(0.0, 0.02)
        self._packed = None
        self._ws = {}
        self.eval()
        self.requires_grad_(False)

    def load_state_dict(self, state_dict, strict=True):
        """Accepts HF CLIPTextModel / CLIPModel checkpoints: an optional ``text_model.`` prefix is stripped,
        ``position_ids`` buffers and non-text entries are dropped."""
        sd = {}
        for k, v in state_dict.items():
            if k.startswith("text_model."):
                k = k[len("text_model."):]
            if k.startswith(("vision_model.", "visual_projection", "text_projection", "logit_scale")) or k.endswith("position_ids"):
                continue
            sd[k] = v
        return super().load_state_dict(sd, strict=strict)

    def _c_cfg(self):
        c = self.cfg
        return _hip.ClipConfig(c["vocab"], c["dim"], c["heads"], c["layers"], c["ffn"], c["max_pos"], c["eps"])

    def invalidate_packed(self):
        """Forget the packed weight blob; needed only after in-place edits through ``p.data`` (same contract as
        UViT.invalidate_packed)."""
        self._packed = None

    def _packed_blob(self, device):
        ps = list(self.parameters())
        versions = tuple((p.data_ptr(), p._version) for p in ps)
        if self._packed is not None and self._packed[0] == device and self._packed[1] == versions:
            return self._packed[2]
        L = _hip.lib()
        cfg = self._c_cfg()
        n = L.uspace_clip_num_params(ctypes.byref(cfg))
        if n != len(ps):
            raise _hip.UspaceHipError(f"CLIP parameter count mismatch: module {len(ps)} vs library {n}")
        srcs = []
        for i, p in enumerate(ps):
            _hip.require_device(p, "parameter")
            if p.numel() != L.uspace_clip_param_numel(ctypes.byref(cfg), i):
                raise _hip.UspaceHipError(f"CLIP parameter {i}: unexpected size {tuple(p.shape)}")
            srcs.append(p.detach().to(torch.float32).contiguous())
        nbytes = L.uspace_clip_weight_bytes(ctypes.byref(cfg))
        blob = torch.empty(nbytes, dtype=torch.uint8, device=device)
        arr = (ctypes.c_void_p * n)(*[t.data_ptr() for t in srcs])
        _hip.check(L.uspace_clip_pack_weights(ctypes.byref(cfg), arr, n, _hip.ptr(blob), nbytes, _hip.stream_ptr()),
                   "uspace_clip_pack_weights")
        torch.cuda.current_stream().synchronize()
        self._packed = (device, versions, blob)
        return blob

    def forward(self, input_ids, hidden_state=None):
        """input_ids [B, L<=max_pos] integer tensor -> last_hidden_state [B, L, D] fp32 (``hidden_state=k``: the state
        after k layers, HF ``output_hidden_states[k]``)."""
        _hip.require_device(input_ids, "input_ids")
        if input_ids.dim() != 2 or input_ids.shape[1] > self.cfg["max_pos"]:
            raise ValueError(f"input_ids must be [B, L<={self.cfg['max_pos']}], got {tuple(input_ids.shape)}")
        if input_ids.numel() and (int(input_ids.min()) < 0 or int(input_ids.max()) >= self.cfg["vocab"]):
            raise IndexError("token id out of range")            # nn.Embedding raises IndexError as well
        dev = input_ids.device
        if input_ids.shape[0] == 0 or input_ids.shape[1] == 0:
            return torch.empty(input_ids.shape[0], input_ids.shape[1], self.cfg["dim"], dtype=torch.float32, device=dev)
        blob = self._packed_blob(dev)
        L = _hip.lib()
        cfg = self._c_cfg()
        B, T = input_ids.shape
        key = (B, str(dev))
        if key not in self._ws:
            self._ws = {key: torch.empty(L.uspace_clip_workspace_bytes(ctypes.byref(cfg), B), dtype=torch.uint8, device=dev)}
        ws = self._ws[key]
        ids = input_ids.to(torch.int32).contiguous()
        out = torch.empty(B, T, self.cfg["dim"], dtype=torch.float32, device=dev)
        _hip.check(L.uspace_clip_text_forward(ctypes.byref(cfg), _hip.ptr(blob), _hip.ptr(ws), ws.numel(), _hip.ptr(ids),
                                              _hip.ptr(out), B, T, -1 if hidden_state is None else int(hidden_state),
                                              _hip.stream_ptr()), "uspace_clip_text_forward")
        return out


class AbstractEncoder(nn.Module):
    def encode(self, *args, **kwargs):
        raise NotImplementedError


class FrozenCLIPEmbedder(AbstractEncoder):
    """``FrozenCLIPEmbedder(version, device, max_length)`` as in libs/clip.py:40-91.  ``tokenizer`` / ``transformer`` may
    be passed in (offline use, tests); otherwise they are loaded with ``from_pretrained(version)`` like the reference
    (which needs the HF files on disk)."""

    def __init__(self, version="openai/clip-vit-large-patch14", device="cuda", max_length=77, tokenizer=None,
                 transformer=None):
        super().__init__()
        if tokenizer is None:
            from transformers import CLIPTokenizer
            tokenizer = CLIPTokenizer.from_pretrained(version)
        if transformer is None:
            from transformers import CLIPTextModel
            hf = CLIPTextModel.from_pretrained(version)
            c = hf.config
            transformer = CLIPTextTransformer(c.vocab_size, c.hidden_size, c.intermediate_size, c.num_hidden_layers,
                                              c.num_attention_heads, c.max_position_embeddings, c.layer_norm_eps, c.hidden_act)
            transformer.load_state_dict(hf.state_dict())
        self.tokenizer = tokenizer
        self.transformer = transformer
        self.device = device
        self.max_length = max_length
        self.freeze()

    def freeze(self):
        self.transformer = self.transformer.eval()
        for p in self.parameters():
            p.requires_grad = False

    def get_word_inds(self, text, word_place):
        return get_word_inds(text=text, word_place=word_place, tokenizer=self.tokenizer)

    def forward(self, text):
        enc = self.tokenizer(text, truncation=True, max_length=self.max_length, return_length=True,
                             return_overflowing_tokens=False, padding="max_length", return_tensors="pt")
        tokens = enc["input_ids"].to(self.device)
        return self.transformer(tokens)

    def encode(self, text):
        return self(text)
