"""Persistent HIP LSTM against torch.nn.LSTM (CPU, fp32) on bf16-quantised operands."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("U,B,H", [(7, 5, 64), (23, 32, 128), (106, 32, 640), (9, 40, 64)])
def test_lstm_forward_backward_match_nn_lstm(U, B, H):
    from indic_cl_asr_amd.ops.lstm import lstm_forward
    torch.manual_seed(U + B)
    ref = torch.nn.LSTM(H, H, 1)
    with torch.no_grad():
        for p in ref.parameters():
            p.copy_(p.bfloat16().float() if p.dim() == 2 else p)   # weights exactly representable in bf16
    x = (torch.randn(U, B, H) * 0.7).bfloat16().float()
    xr = x.clone().requires_grad_(True)
    y_ref, _ = ref(xr)
    gy = torch.randn(U, B, H)
    y_ref.backward(gy)
    m = torch.nn.LSTM(H, H, 1).cuda()
    m.load_state_dict(ref.state_dict())
    xc = x.cuda().requires_grad_(True)
    y = lstm_forward(xc, m)
    y.backward(gy.cuda())
    torch.cuda.synchronize()
    assert (y.cpu() - y_ref).abs().max().item() < 3e-2       # bf16 hand-off of h_t between steps
    sc = lambda t: t.abs().max().item()
    assert (xc.grad.cpu() - xr.grad).abs().max().item() < 4e-2 * sc(xr.grad) + 1e-5
    for n, p in m.named_parameters():
        g = dict(ref.named_parameters())[n].grad
        assert (p.grad.cpu() - g).abs().max().item() < 4e-2 * sc(g) + 1e-5, n
