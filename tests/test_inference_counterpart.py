"""The inference caller counterpart (reference inference_wavemamba.py:28-36, :99-113): pad to x128 ->
restoration_network -> crop -> uint8 quantisation -> PSNR.  CPU: oracle ops backend (test infrastructure);
GPU: HIP path, compared with the CPU-oracle result."""
import pytest
import torch

from oracle import oracle
from oracle import backend as oracle_backend
import wave_mamba_amd as wm
from wave_mamba_amd.archs import wavemamba_arch as arch

CFG = dict(in_chn=3, wf=8, n_l_blocks=[1, 1, 1], n_h_blocks=[1, 1, 1], ffn_scale=2.0)


def run_cpu(net, img):
    prev = oracle_backend.set_ops_backend(oracle)
    try:
        return wm.inference.enhance(net, img)
    finally:
        oracle_backend.set_ops_backend(prev)


def test_pad_crop_quantise_psnr_cpu():
    torch.manual_seed(0)
    net = wm.WaveMamba(**CFG).eval()
    img = torch.rand(1, 3, 70, 150, generator=torch.Generator().manual_seed(1234))
    padded = wm.inference.check_image_size(img)
    assert padded.shape == (1, 3, 128, 256)
    assert torch.equal(padded[:, :, :70, :150], img)
    assert torch.equal(padded[:, :, 70:75, :150], img[:, :, [68, 67, 66, 65, 64], :])     # reflect, no edge repeat
    out = run_cpu(net, img)
    assert out.shape == img.shape and not out.requires_grad
    q = wm.inference.to_uint8(out)
    assert q.dtype == torch.uint8
    tgt = wm.inference.to_uint8(torch.rand(1, 3, 70, 150, generator=torch.Generator().manual_seed(4321)))
    p = wm.inference.psnr_uint8(q, tgt)
    assert 3.0 < p < 30.0
    assert wm.inference.psnr_uint8(q, q) == float("inf")
    # formula check against a hand computation
    a = torch.tensor([[0, 255], [10, 20]], dtype=torch.uint8)
    b = torch.tensor([[0, 250], [12, 20]], dtype=torch.uint8)
    mse = (25 + 4) / 4
    assert abs(wm.inference.psnr_uint8(a, b) - 20 * torch.log10(torch.tensor(255.0 / mse ** 0.5)).item()) < 1e-6


@pytest.mark.gpu
def test_enhance_gpu_matches_cpu_oracle_psnr():
    torch.manual_seed(0)
    net = wm.WaveMamba(**CFG).eval()
    img = torch.rand(1, 3, 200, 328, generator=torch.Generator().manual_seed(1234))
    ref = run_cpu(net, img)
    got = wm.inference.enhance(net.to("cuda:0"), img.to("cuda:0")).cpu()
    assert float((got - ref).norm() / ref.norm()) <= 1e-4
    tgt = wm.inference.to_uint8(torch.rand(img.shape, generator=torch.Generator().manual_seed(4321)))
    d = abs(wm.inference.psnr_uint8(wm.inference.to_uint8(got), tgt) - wm.inference.psnr_uint8(wm.inference.to_uint8(ref), tgt))
    assert d <= 1e-3            # BASELINE north_star: within 1e-3 dB PSNR of the reference CPU path
