"""TEST INFRASTRUCTURE ONLY - CPU restatement of the embedding splice of the reference,
`LlavaMetaForCausalLM.prepare_inputs_labels_for_multimodal` (`llava/model/llava_arch.py:210-332`: everything after the image
features exist).  Plain Python loops, one sample at a time, like the reference.  Pinned by tests/test_splice.py against the
reference method itself (when `/root/reference` is mounted) on random batches covering every branch."""
from __future__ import annotations

from typing import List, Optional

import torch

IGNORE_INDEX = -100          # llava/constants.py:7
IMAGE_TOKEN_INDEX = -200     # llava/constants.py:8


def splice(input_ids, position_ids, attention_mask, labels, image_features: List[torch.Tensor], embed_weight,
           padding_side: str = "right", max_length: Optional[int] = None):
    _labels, _position_ids, _attention_mask = labels, position_ids, attention_mask            # :218-220
    if attention_mask is None:
        attention_mask = torch.ones_like(input_ids, dtype=torch.bool)
    else:
        attention_mask = attention_mask.bool()
    if position_ids is None:
        position_ids = torch.arange(0, input_ids.shape[1], dtype=torch.long)
    if labels is None:
        labels = torch.full_like(input_ids, IGNORE_INDEX)
    ids = [i[m] for i, m in zip(input_ids, attention_mask)]                                     # :228-231
    labs = [l[m] for l, m in zip(labels, attention_mask)]
    embeds, new_labels, k = [], [], 0
    for b, cur in enumerate(ids):                                                               # :236-290
        n_img = int((cur == IMAGE_TOKEN_INDEX).sum())
        if n_img == 0:
            embeds.append(torch.cat([embed_weight[cur], image_features[k][0:0]], 0))
            new_labels.append(labs[b])
            k += 1
            continue
        cuts = [-1] + torch.where(cur == IMAGE_TOKEN_INDEX)[0].tolist() + [cur.shape[0]]
        pe, pl = [], []
        for i in range(len(cuts) - 1):
            seg = cur[cuts[i] + 1:cuts[i + 1]]
            pe.append(embed_weight[seg])
            pl.append(labs[b][cuts[i] + 1:cuts[i + 1]])
            if i < n_img:
                f = image_features[k]
                k += 1
                pe.append(f)
                pl.append(torch.full((f.shape[0],), IGNORE_INDEX, dtype=labs[b].dtype))
        embeds.append(torch.cat(pe, 0))
        new_labels.append(torch.cat(pl, 0))
    if max_length is not None:                                                                  # :292-296
        embeds = [e[:max_length] for e in embeds]
        new_labels = [l[:max_length] for l in new_labels]
    max_len = max(e.shape[0] for e in embeds)                                                   # :299
    B, H = len(embeds), embed_weight.shape[1]
    out = torch.zeros(B, max_len, H, dtype=embed_weight.dtype)
    lab = torch.full((B, max_len), IGNORE_INDEX, dtype=new_labels[0].dtype)
    am = torch.zeros(B, max_len, dtype=attention_mask.dtype)
    pos = torch.zeros(B, max_len, dtype=position_ids.dtype)
    for i, (e, l) in enumerate(zip(embeds, new_labels)):                                        # :304-323
        n = e.shape[0]
        if n == 0:
            continue
        sl = slice(max_len - n, max_len) if padding_side == "left" else slice(0, n)
        out[i, sl], lab[i, sl], am[i, sl] = e, l, True
        pos[i, sl] = torch.arange(0, n, dtype=pos.dtype)
    return (None, None if _position_ids is None else pos, None if _attention_mask is None else am.to(_attention_mask.dtype), None, out,
            None if _labels is None else lab)
