from .rnnt import RNNTLossHIP, RNNTLoss, rnnt_loss_hip  # noqa: F401
