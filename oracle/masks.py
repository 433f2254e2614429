"""CPU restatement of DCFPPruner's threshold / mask arithmetic.
Follows pruners/dcfp_pruner.py:36-37 (groups), :43-66 (get_thresh), :68-92 (gen_channel_mask).
torch.sort on CPU is used where the reference uses it (tie order of the forced top-k)."""
import torch


def group_of(bn_name):
    return 0 if bn_name.startswith("backbone") else 1


def thresholds(scores, bn_names, except_layers, global_percent):
    """scores: {bn_name: FloatTensor[C]}; bn_names: the BN layers linked to a conv, any order."""
    pools = [[], []]
    for bn in bn_names:
        if bn not in except_layers:
            pools[group_of(bn)].append(scores[bn].float())
    th = [0, 0]
    for g in range(2):
        if pools[g]:
            allv = torch.cat(pools[g])
            th[g] = torch.sort(allv)[0][int(allv.numel() * global_percent)]
    return th


def out_masks(scores, norm_conv_links, except_layers, th, layer_keep):
    """{conv_name: FloatTensor[C] of 0/1} for every conv whose name is not excepted."""
    masks = {}
    for bn, conv in norm_conv_links.items():
        if conv in except_layers:
            continue
        s = scores[bn]
        C = s.numel()
        m = s.gt(th[group_of(bn)]).float()
        k = int(C * layer_keep) if int(C * layer_keep) > 0 else 1
        if int(m.sum()) < k:
            order = torch.sort(s, descending=True)[1]
            m[order[:k]] = 1.0
        masks[conv] = m
    return masks
