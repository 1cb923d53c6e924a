"""CTC loss with the reference's wrapper semantics (A/losses/ctc.py:45-82): blank = num_classes,
zero_infinity, 'mean_batch' = mean over the per-utterance losses."""
import torch
import torch.nn as nn


class CTCLoss(nn.Module):
    def __init__(self, num_classes, zero_infinity=False, reduction='mean_batch'):
        super().__init__()
        if reduction not in ['none', 'mean', 'sum', 'mean_batch', 'mean_volume']:
            raise ValueError('`reduction` must be one of [mean, sum, mean_batch, mean_volume]')
        self._blank = num_classes
        self.zero_infinity = zero_infinity
        self.config_reduction = reduction
        self._apply_reduction = reduction in ('mean_batch', 'mean_volume')
        self._ctc_reduction = 'none' if self._apply_reduction else reduction

    def reduce(self, losses, target_lengths):
        if self.config_reduction == 'mean_batch':
            losses = losses.mean()
        elif self.config_reduction == 'mean_volume':
            losses = losses.sum() / target_lengths.sum()
        return losses

    def forward(self, log_probs, targets, input_lengths, target_lengths):
        input_lengths = input_lengths.long()
        target_lengths = target_lengths.long()
        targets = targets.long()
        log_probs = log_probs.transpose(1, 0)  # [B,T,D] -> [T,B,D]
        loss = torch.nn.functional.ctc_loss(log_probs, targets, input_lengths, target_lengths, blank=self._blank,
                                            reduction=self._ctc_reduction, zero_infinity=self.zero_infinity)
        if self._apply_reduction:
            loss = self.reduce(loss, target_lengths)
        return loss
