"""CPU restatement of the evaluation metrics.  Follows evaluate.py:229-247 (confusion matrix via
bincount over gt*C+pred) and :374-380 (IoU = tp / (pos + res - tp), mean over classes)."""
import numpy as np


def confusion_matrix(gt_label, pred_label, class_num):
    index = (gt_label * class_num + pred_label).astype("int32")
    label_count = np.bincount(index)
    cm = np.zeros((class_num, class_num))
    for i in range(class_num):
        for j in range(class_num):
            cur = i * class_num + j
            if cur < len(label_count):
                cm[i, j] = label_count[cur]
    return cm


def mean_iou(cm):
    pos, res, tp = cm.sum(1), cm.sum(0), np.diag(cm)
    iou = tp / np.maximum(1.0, pos + res - tp)
    return iou.mean(), iou
