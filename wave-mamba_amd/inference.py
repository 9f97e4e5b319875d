"""Counterpart of the reference's inference caller for the hot path (inference_wavemamba.py:28-36, :99-113)
and of its image <-> tensor conventions (basicsr/utils/img_util.py:67-94) and PSNR
(comput_psnr_ssim.py:434-438), on tensors instead of image files (no cv2 / datasets in scope).

    pad (reflect, to a multiple of 128)  ->  EnhanceNet.restoration_network(x) under no_grad
    ->  crop back to (h, w)  ->  clamp [0,1] * 255, round  ->  uint8  ->  PSNR against a target
"""
import torch
import torch.nn.functional as F


def check_image_size(x, window_size=128):
    """inference_wavemamba.py:28-36: reflect-pad bottom/right so H and W are multiples of window_size."""
    _, _, h, w = x.shape
    pad_h = (window_size - h % window_size) % window_size
    pad_w = (window_size - w % window_size) % window_size
    return F.pad(x, (0, pad_w, 0, pad_h), "reflect")


@torch.no_grad()
def enhance(net, img, window_size=128):
    """img: (B, 3, h, w) float in [0, 1] on the model's device -> restored image, same shape
    (inference_wavemamba.py:99-113: pad -> restoration_network -> crop)."""
    _, _, h, w = img.shape
    out = net.restoration_network(check_image_size(img, window_size))
    return out[:, :, :h, :w]


def to_uint8(t):
    """tensor2img's quantisation (img_util.py:67-94, min_max=(0,1)): clamp, scale by 255, round."""
    return (t.detach().float().clamp(0, 1) * 255.0).round().to(torch.uint8)


def psnr_uint8(a, b):
    """20*log10(255/sqrt(mse)) on uint8-valued images (comput_psnr_ssim.py:434-438, crop_border 0, RGB)."""
    mse = (a.double() - b.double()).pow(2).mean()
    if float(mse) == 0.0:
        return float("inf")
    return float(20.0 * torch.log10(255.0 / mse.sqrt()))
