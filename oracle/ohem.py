"""CPU restatement of OHEM's threshold search and relabelling, numpy + scipy like the
reference.  Follows loss/ohem.py:20-48 (find_threshold) and :51-78 (generate_new_target)."""
import numpy as np
import scipy.ndimage as nd


def find_threshold(np_predict, np_target, ignore_label=255, thresh=0.7, min_kept=100000, factor=8):
    predict = nd.zoom(np_predict, (1.0, 1.0, 1.0 / factor, 1.0 / factor), order=1)
    target = nd.zoom(np_target, (1.0, 1.0 / factor, 1.0 / factor), order=0)
    n, c, h, w = predict.shape
    min_kept = min_kept // (factor * factor)
    input_label = target.ravel().astype(np.int32)
    input_prob = np.rollaxis(predict, 1).reshape((c, -1))
    valid_flag = input_label != ignore_label
    label = input_label[valid_flag]
    num_valid = valid_flag.sum()
    threshold = None
    if min_kept >= num_valid:
        threshold = 1.0
    elif num_valid > 0:
        prob = input_prob[:, valid_flag]
        pred = prob[label, np.arange(len(label), dtype=np.int32)]
        threshold = thresh
        if min_kept > 0:
            k_th = min(len(pred), min_kept) - 1
            new_threshold = np.partition(pred, k_th)[k_th]
            if new_threshold > thresh:
                threshold = new_threshold
    return threshold


def new_target(np_predict, np_target, ignore_label=255, thresh=0.7, min_kept=100000, factor=8):
    n, c, h, w = np_predict.shape
    threshold = find_threshold(np_predict, np_target, ignore_label, thresh, min_kept, factor)
    input_label = np_target.ravel().astype(np.int32)
    input_prob = np.rollaxis(np_predict, 1).reshape((c, -1))
    valid_flag = input_label != ignore_label
    valid_inds = np.where(valid_flag)[0]
    label = input_label[valid_flag]
    if valid_flag.sum() > 0:
        prob = input_prob[:, valid_flag]
        pred = prob[label, np.arange(len(label), dtype=np.int32)]
        valid_inds = valid_inds[pred <= threshold]
    label = input_label[valid_inds].copy()
    input_label.fill(ignore_label)
    input_label[valid_inds] = label
    return input_label.reshape(np_target.shape), threshold
