"""Host-side logic of the T2I attention-map edit (reference: tools/utils_t2i.py:196-296).

The reference materialises softmax(QK^T) [B,H,L,L], multiplies the columns of the target
context tokens, and multiplies by V.  Because the edit happens after softmax with no
renormalisation, (P * colscale) @ V == P @ (diag(colscale) V): the HIP attention kernel takes
a per-(batch, key) factor table instead, and never materialises the map.  This file only
decides WHICH factors apply at a given step / block (same control flow as the reference).
"""
import numpy as np

TIME_TOKEN_NUM = 1
EDIT_NAMES = ("p2p", "local_prompt", "sampled_image_editing")


def uses_attention_edit_path(kwargs):
    """libs/uvit_t2i.py:91-96: these dissect names route attention through the editable map."""
    return kwargs.get("dissect_name") in EDIT_NAMES


def block_selected(target_block_id, block_id):
    """tools/utils_t2i.py:227-238."""
    if isinstance(target_block_id, (int, np.integer)) and not isinstance(target_block_id, bool):
        return block_id == int(target_block_id)
    if isinstance(target_block_id, (list, tuple)):
        return block_id in target_block_id
    if isinstance(target_block_id, str) and target_block_id == "all":
        return True
    if target_block_id is None:
        return True
    raise ValueError(f"unknown target_block_id {target_block_id}")


def key_scale_table(n_blocks, B, L, timestep_digit, kwargs):
    """[n_blocks, B, L] fp32 factors (1 = untouched) or None when this step edits nothing.

    Error behaviour follows the reference: ValueError for a foreign dissect_name,
    NotImplementedError for an unknown fm_direction / token_dissect."""
    name = kwargs.get("dissect_name")
    if name not in EDIT_NAMES:
        raise ValueError(f"dissect_name should be read or write, here is {name}")
    direction = kwargs.get("fm_direction")
    if direction == "encode":
        return None
    if direction != "decode":
        raise NotImplementedError(f"fm_direction={direction}")
    if not float(timestep_digit) <= kwargs.get("t_edit"):
        return None
    tk = kwargs["token_kwargs"]
    mode = tk["token_dissect"]
    if mode.startswith("lp_"):
        return None
    if mode != "p2p_rescale":
        raise NotImplementedError(mode)
    ids = kwargs["target_context_ids"]
    mult = tk["p2p_multiplier"]
    if isinstance(mult, (int, float)):
        mult = [mult] * len(ids)
    elif not isinstance(mult, list):
        raise ValueError(f"unknown p2p_multiplier {mult}")
    row = np.ones((B, L), np.float32)
    touched = False
    for b, tid in enumerate(ids):
        tid = np.asarray(tid)
        if tid.size > 0:
            row[b, tid.astype(np.int64) + TIME_TOKEN_NUM] = np.float32(mult[b])
            touched = True
    table = np.ones((n_blocks, B, L), np.float32)
    any_block = False
    for blk in range(n_blocks):
        if block_selected(kwargs.get("block_id"), blk):
            table[blk] = row
            any_block = True
    if not (touched and any_block):
        return None
    return table
