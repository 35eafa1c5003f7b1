"""Drop-in for graphattention/BPRLoss.py:4-9: BPRLoss().forward(pos_scores, neg_scores)."""
from torch import nn

from . import functional as NF


class BPRLoss(nn.Module):
    def __init__(self):
        super().__init__()

    def forward(self, pos_scores, neg_scores):
        return NF.bpr_loss(pos_scores, neg_scores)
