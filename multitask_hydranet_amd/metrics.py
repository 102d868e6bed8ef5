"""Streaming segmentation IoU on the device (reference: head_seg/seg_metrics.py:12-101 stat_scores_multiple_classes + IntersectionOverUnion,
used by train.py's validation loop).  update() adds one batch to a (C+1) x (C+1) uint64 confusion matrix with ONE kernel launch (integer
atomics: exact and order independent; the reference accumulates float32 counters, which stop being exact above 2^24 pixels per class);
compute() derives the per-class scores with the reference's rules (absent_score, ignore_index removal)."""
from __future__ import annotations

from typing import Optional

import torch

from ._lib import lib


class IntersectionOverUnion:
    def __init__(self, n_classes: int, ignore_index: Optional[int] = None, absent_score: float = 0.0, reduction: str = "none", device=None):
        self.n_classes, self.ignore_index, self.absent_score, self.reduction = n_classes, ignore_index, absent_score, reduction
        self.device = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())
        self.conf = torch.zeros(((n_classes + 1) ** 2,), device=self.device, dtype=torch.int64)

    def update(self, prediction: torch.Tensor, target: torch.Tensor):
        pred = prediction.to(self.device)
        pred = pred.long().contiguous().view(-1)
        tgt = target.to(self.device).contiguous().view(-1)
        if tgt.dtype not in (torch.int64, torch.float32):
            tgt = tgt.long()
        assert pred.numel() == tgt.numel()
        lib().call("hn_seg_confusion", pred.data_ptr(), tgt.data_ptr(), 1 if tgt.dtype == torch.float32 else 0, pred.numel(), self.n_classes,
                   self.conf.data_ptr())

    def stats(self):
        """(true_positive, false_positive, false_negative, support) per class, int64"""
        c = self.n_classes
        m = self.conf.view(c + 1, c + 1)                       # m[pred][target]
        tp = torch.diagonal(m)[:c]
        fp = m.sum(1)[:c] - tp
        fn = m.sum(0)[:c] - tp
        return tp, fp, fn, m.sum(0)[:c]

    def compute(self):
        tp, fp, fn, sup = (t.to(torch.float32) for t in self.stats())
        scores = torch.zeros(self.n_classes, device=self.device, dtype=torch.float32)
        for k in range(self.n_classes):
            if k == self.ignore_index:
                continue
            if float(sup[k] + tp[k] + fp[k]) == 0:
                scores[k] = self.absent_score
                continue
            scores[k] = tp[k] / (tp[k] + fp[k] + fn[k])
        if self.ignore_index is not None and 0 <= self.ignore_index < self.n_classes:
            scores = torch.cat([scores[:self.ignore_index], scores[self.ignore_index + 1:]])
        return scores
