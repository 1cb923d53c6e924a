"""Conformer convolution-module core: GLU -> pad mask -> depthwise conv k -> BatchNorm (batch stats incl. the
zeroed padded frames, as the reference) -> SiLU  (conformer_modules.py:340-366)."""
import torch
import torch.nn.functional as F


def glu_dwconv_bn_silu(x2, pad_mask, dw_weight, dw_bias, bn, training):
    """x2: [B,2d,T] output of pointwise_conv1; pad_mask [B,T] True at padding; bn: nn.BatchNorm1d (running stats
    updated in train mode).  Returns [B,d,T] f32."""
    x = F.glu(x2, dim=1)
    x = x.float().masked_fill(pad_mask.unsqueeze(1), 0.0)
    k = dw_weight.shape[-1]
    # fp32 grouped conv outside autocast: in bf16 MIOpen falls back to its naive depthwise kernels on gfx950
    # (1.6 ms per weight-gradient at bs32 x 15 s); the reference runs this op in fp32 as well (x.float(), :351)
    with torch.autocast(device_type=x.device.type, enabled=False):
        x = F.conv1d(F.pad(x.float(), ((k - 1) // 2, (k - 1) // 2)), dw_weight.float(), dw_bias.float(), groups=x.shape[1])
    x = bn(x)  # module call: survives SyncBatchNorm.convert_sync_batchnorm (R/cl_baseline.py:133)
    return F.silu(x)
