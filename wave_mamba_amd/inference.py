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


# ------------------------------------------------------------------------------------------------------------------
# uint8 host images in, uint8 host images out (inference_wavemamba.py:99-113 end to end):
# pinned double-buffered H2D / D2H on side streams under the previous / next image's forward.
# ------------------------------------------------------------------------------------------------------------------
class UInt8Pipeline:
    """enhance host uint8 images (h, w, 3; BGR as cv2 reads them, or RGB with swap_rb=False) with the network on `device`.

    Per image: pinned staging buffer -> H2D on the upload stream -> wm_image_pre_u8 (CHW, / 255, reflect pad) ->
    restoration_network -> wm_image_post_u8 (crop, clamp, * 255, round, HWC) -> D2H on the download stream into a pinned
    buffer.  Two buffers each way, events between the streams: the copies of images i + 1 / i - 1 run under the forward
    of image i.  `run(images)` yields numpy uint8 results in order."""

    def __init__(self, net, device, window_size=128, swap_rb=True):
        from . import ops
        self.net, self.device, self.window, self.swap_rb, self.ops = net, torch.device(device), window_size, swap_rb, ops
        self.up, self.down = torch.cuda.Stream(self.device), torch.cuda.Stream(self.device)
        self._pin_in, self._pin_out, self._dev_in = [None, None], [None, None], [None, None]
        self._up_done = [None, None]       # per slot: the last H2D copy out of pin_in[slot] (the host may then refill it)
        self._consumed = [None, None]      # per slot: main-stream event after image_pre_u8 has read dev_in[slot]

    def _buf(self, store, slot, shape, pinned):
        t = store[slot]
        if t is None or tuple(t.shape) != tuple(shape):
            t = (torch.empty(shape, dtype=torch.uint8, pin_memory=True) if pinned
                 else torch.empty(shape, dtype=torch.uint8, device=self.device))
            store[slot] = t
        return t

    @torch.no_grad()
    def run(self, images):
        import numpy as np
        main = torch.cuda.current_stream(self.device)
        pending = []                                   # (pinned output, download-done event)
        uploaded = None
        it = iter(images)

        def upload(img, slot):
            """Stage `img` into slot: host -> pinned -> device on the upload stream.  Orderings (all cross-stream):
              * the host refills pin_in[slot] only after the previous H2D copy out of it has finished;
              * the H2D copy into dev_in[slot] waits for main's image_pre_u8 of the previous image of this slot;
              * the device buffer is allocated with `up` current (its pool), waits for main when it is (re)allocated -
                a fresh block may be memory that kernels still queued on main are using - and is recorded on main,
                which reads it."""
            a = torch.from_numpy(np.ascontiguousarray(img))
            if self._up_done[slot] is not None:
                self._up_done[slot].synchronize()
            pin = self._buf(self._pin_in, slot, a.shape, True)
            pin.copy_(a)
            with torch.cuda.stream(self.up):
                if self._consumed[slot] is not None:
                    self.up.wait_event(self._consumed[slot])
                old = self._dev_in[slot]
                if old is None or tuple(old.shape) != tuple(a.shape):
                    self.up.wait_stream(main)
                dev = self._buf(self._dev_in, slot, a.shape, False)
                dev.copy_(pin, non_blocking=True)
                ev = torch.cuda.Event(); ev.record(self.up)
            dev.record_stream(main)
            self._up_done[slot] = ev
            return dev, ev

        nxt = next(it, None)
        slot = 0
        if nxt is not None:
            uploaded = upload(nxt, slot)
        while uploaded is not None:
            dev, ev = uploaded
            nxt = next(it, None)
            uploaded = upload(nxt, slot ^ 1) if nxt is not None else None     # overlaps with the forward below
            main.wait_event(ev)
            h, w = dev.shape[:2]
            x = self.ops.image_pre_u8(dev, self.window, self.swap_rb)
            c = torch.cuda.Event(); c.record(main)
            self._consumed[slot] = c                   # dev_in[slot] may be overwritten by the upload stream after this
            y = self.net.restoration_network(x)
            res = self.ops.image_post_u8(y, h, w, self.swap_rb)
            done = torch.cuda.Event(); done.record(main)
            pin_out = self._buf(self._pin_out, slot, res.shape, True)
            with torch.cuda.stream(self.down):
                self.down.wait_event(done)
                pin_out.copy_(res, non_blocking=True)
                res.record_stream(self.down)
                dl = torch.cuda.Event(); dl.record(self.down)
            pending.append((pin_out, dl))
            if len(pending) == 2:                      # the slot about to be reused must have been handed out
                p, e = pending.pop(0)
                e.synchronize()
                yield p.numpy().copy()
            slot ^= 1
        for p, e in pending:
            e.synchronize()
            yield p.numpy().copy()


# ------------------------------------------------------------------------------------------------------------------
# bf16-storage mode (BASELINE config 2 as worded; ops.set_plane_dtype)
# ------------------------------------------------------------------------------------------------------------------
@torch.no_grad()
def bench_bf16_storage(net, x, steps, warmup=2):
    """Time `steps` forwards with the LFSSBlock-internal planes stored as bfloat16 (fp32 arithmetic and state inside
    every kernel, fp32 tokens between blocks, fp32 HFE branch) and compare with the fp32 forward of the same input:
    PSNR on [0, 1]-clamped float outputs, PSNR after the reference's uint8 quantisation, relative l2."""
    import time
    from . import ops
    prev = ops.set_plane_dtype(torch.float32)
    try:
        ref = net.restoration_network(x)
        ops.set_plane_dtype(torch.bfloat16)
        for _ in range(max(warmup, 1)):
            out = net.restoration_network(x)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            out = net.restoration_network(x)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    finally:
        ops.set_plane_dtype(prev)
    mse = float((out.clamp(0, 1) - ref.clamp(0, 1)).double().pow(2).mean())
    return {"images_per_s": steps / dt, "ms_per_step": 1e3 * dt / steps,
            "psnr_vs_fp32_db": float("inf") if mse == 0 else float(10 * torch.log10(torch.tensor(1.0 / mse))),
            "psnr_u8_vs_fp32_db": psnr_uint8(to_uint8(out), to_uint8(ref)),
            "rel_l2_vs_fp32": float((out - ref).norm() / ref.norm()),
            "note": "bf16 planes between the kernels of every LFSSBlock (x, z, conv outputs, the four scan outputs, f, fc); "
                    "fp32 tiles / projections / scan state / statistics inside the kernels, fp32 tokens, fp32 HFE branch"}
