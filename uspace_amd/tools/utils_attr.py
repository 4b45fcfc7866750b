"""Attribute directions for the u-space write hook, accumulated on the device
(reference: tools/utils_attr.py:124-207, the README's "step 2").

The reference's read hook writes every tapped activation to ``{batch_id}_{t:.2f}.npy``; an offline numpy
pass then reloads all of them ([N, T, ...], ~0.5 TB for 5 000 mid-block samples x 100 steps), and per
attribute takes mean(pos) - mean(neg).  Here the running sums live in HBM: ``DirectionAccumulator.update``
adds one batch of activations for one timestep (one HIP kernel), ``finalize`` divides by the counts and writes
the same ``delta_{t}.npy`` files ([n_attr, ...] fp32) the write hook reads.
"""
import os

import numpy as np
import torch

from .. import _hip


# Attribute tables of the two annotated face datasets, in annotation-file column order (list_attr_celeba.txt; the 11 FFHQ
# columns of the reference's label file).  Only the names used in output file names (tools/utils_attr.py:14-23, 73-85).
CELEBA_ATTRS = [
    "5_o_Clock_Shadow", "Arched_Eyebrows", "Attractive", "Bags_Under_Eyes", "Bald", "Bangs", "Big_Lips", "Big_Nose",
    "Black_Hair", "Blond_Hair", "Blurry", "Brown_Hair", "Bushy_Eyebrows", "Chubby", "Double_Chin", "Eyeglasses", "Goatee",
    "Gray_Hair", "Heavy_Makeup", "High_Cheekbones", "Male", "Mouth_Slightly_Open", "Mustache", "Narrow_Eyes", "No_Beard",
    "Oval_Face", "Pale_Skin", "Pointy_Nose", "Receding_Hairline", "Rosy_Cheeks", "Sideburns", "Smiling", "Straight_Hair",
    "Wavy_Hair", "Wearing_Earrings", "Wearing_Hat", "Wearing_Lipstick", "Wearing_Necklace", "Wearing_Necktie", "Young",
]
FFHQ_ATTRS = ["gender", "smile", "no_glasses", "anger", "contempt", "disgust", "fear", "happiness", "neutral", "sadness",
              "surprise"]


def get_attr_name_from_attr_id(ith_attr, dataset_name):
    """File-name piece for an attribute id or an "a_b_c" id string (tools/utils_attr.py:104-121): names joined by "_";
    ``ValueError`` for an unknown dataset or id type, like the reference."""
    if "ffhq" in dataset_name:
        table = FFHQ_ATTRS
    elif "celeba" in dataset_name:
        table = CELEBA_ATTRS
    else:
        raise ValueError("unknown dataset_name", dataset_name)
    if isinstance(ith_attr, (int, np.integer)) and not isinstance(ith_attr, bool):
        return table[int(ith_attr)]
    if isinstance(ith_attr, str):
        return "_".join(table[int(tok)] for tok in ith_attr.split("_"))
    raise ValueError("unknown ith_attr", ith_attr)


class DirectionAccumulator:
    def __init__(self, attr_dim):
        if attr_dim not in (40, 11):       # CelebA-40 / FFHQ-11, the only tables the reference accepts
            raise ValueError("unknown attr dim", attr_dim)
        self.attr_dim = attr_dim
        self._state = {}                   # timestep digit -> (pos_sum [A,F], neg_sum [A,F], n_pos [A], n_neg [A], shape)

    def update(self, timestep_digit, feats, attrs):
        """feats [B, ...] fp32 CUDA tensor (an activation tapped by the hook), attrs [B, attr_dim] ints."""
        _hip.require_device(feats, "feats")
        B = feats.shape[0]
        attrs_t = torch.as_tensor(np.asarray(attrs.cpu() if torch.is_tensor(attrs) else attrs)).to(torch.int32)
        if attrs_t.shape != (B, self.attr_dim):
            raise ValueError(f"attrs must be [{B},{self.attr_dim}], got {tuple(attrs_t.shape)}")
        f = feats.detach().to(torch.float32).contiguous().view(B, -1)
        F = f.shape[1]
        if F % 4:
            raise ValueError("feature size must be a multiple of 4")
        st = self._state.get(timestep_digit)
        if st is None:
            st = [torch.zeros(self.attr_dim, F, device=f.device), torch.zeros(self.attr_dim, F, device=f.device),
                  np.zeros(self.attr_dim, np.int64), np.zeros(self.attr_dim, np.int64), tuple(feats.shape[1:])]
            self._state[timestep_digit] = st
        elif st[4] != tuple(feats.shape[1:]):
            raise ValueError("feature shape changed between updates")
        a_dev = attrs_t.to(f.device).contiguous()
        _hip.check(_hip.lib().uspace_direction_accumulate(_hip.ptr(f), _hip.ptr(a_dev), _hip.ptr(st[0]), _hip.ptr(st[1]),
                                                          B, F, self.attr_dim, _hip.stream_ptr()),
                   "uspace_direction_accumulate")
        an = attrs_t.numpy()
        st[2] += (an == 1).sum(axis=0)
        st[3] += (an == 0).sum(axis=0)
        torch.cuda.current_stream().synchronize()      # a_dev / f may be temporaries

    def directions(self, timestep_digit):
        """[attr_dim, ...] fp32 numpy: mean(pos) - mean(neg); NaN where a side has no example (as numpy's
        empty mean in the reference)."""
        pos, neg, n_pos, n_neg, shape = self._state[timestep_digit]
        with np.errstate(invalid="ignore", divide="ignore"):
            p = pos.cpu().numpy() / n_pos.astype(np.float32)[:, None]
            q = neg.cpu().numpy() / n_neg.astype(np.float32)[:, None]
        return (p - q).astype(np.float32).reshape((self.attr_dim,) + shape)

    def finalize(self, write_root):
        """Write ``delta_{t}.npy`` per timestep (tools/utils_attr.py:201-206)."""
        os.makedirs(write_root, exist_ok=True)
        for t in sorted(self._state):
            np.save(os.path.join(write_root, f"delta_{t}"), self.directions(t))
        return sorted(self._state)
