"""Aligned-face serving pipeline: PNG files -> restored PNG files with the host work overlapped with the GPU.

The reference restores one face per forward and pays, per face, two host passes over the pixels (img2tensor + normalize,
tensor2img), a blocking device->host copy and `torch.cuda.empty_cache()` (inference_codeformer.py:197-206).  Here a batch moves as
    decode (worker pool) -> pinned uint8 staging -> H2D on a copy stream -> cf_img_u8_to_tensor -> CodeFormer.forward ->
    cf_tensor_to_img_u8 -> D2H on a second copy stream into pinned uint8 staging -> encode + write (worker pool)
with a ring of staging slots, so that while batch k is on the GPU the pool already decodes batch k+1.. and still encodes batch k-1.
Only uint8 crosses PCIe (0.79 MB per face each way instead of 3.1 MB of fp32) and the compute stream never waits for the host.

All CUDA work is issued from the calling thread; workers touch numpy / PIL only (and wait on a CUDA event before reading a slot).
"""
import os
import time
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import torch

from .utils.img_util import imread_bgr, imwrite, resize_bilinear


class _Slot:
    def __init__(self, batch, size, device):
        self.pin_in = torch.empty((batch, size, size, 3), dtype=torch.uint8).pin_memory()
        self.pin_out = torch.empty((batch, size, size, 3), dtype=torch.uint8).pin_memory()
        self.dev_in = torch.empty((batch, size, size, 3), dtype=torch.uint8, device=device)
        self.dev_out = torch.empty((batch, size, size, 3), dtype=torch.uint8, device=device)
        self.ev_in = torch.cuda.Event()
        self.ev_done = torch.cuda.Event()
        self.ev_out = torch.cuda.Event()
        self.writes = []      # futures of the encode tasks still reading pin_out
        self.meta = None


class AlignedFacePipeline:
    """restore(paths, out_paths): every input PNG (aligned crop, resized to 512x512 when needed) -> restored PNG.

    net:      a CodeFormer module on a ROCm device (anything with the reference call signature net(x, w=, adain=) -> (out, ...))
    post:     optional callable(face_bgr_u8, restored_bgr_u8, meta) -> image to write (runs in a worker; default: the restored face)
    Returns a dict of stage timings; `failures` counts faces that fell back to their input (the reference's behaviour on an
    inference error, inference_codeformer.py:207-209) unless strict=True.
    """

    def __init__(self, net, device, batch_size=16, workers=None, slots=4, size=512, png_compress_level=3):
        self.net, self.device, self.batch, self.size = net, torch.device(device), int(batch_size), size
        self.workers = workers or min(32, max(4, (os.cpu_count() or 8) // 2))
        self.png_level = png_compress_level
        self.slots = [_Slot(self.batch, size, self.device) for _ in range(max(2, slots))]
        self.h2d = torch.cuda.Stream(device=self.device)
        self.d2h = torch.cuda.Stream(device=self.device)

    # ---- worker-side (no CUDA calls except Event.synchronize) ------------------------------------------------------------------
    def _decode(self, path, dst):
        img = resize_bilinear(imread_bgr(path), (self.size, self.size))
        dst.copy_(torch.from_numpy(img))
        return img.shape

    def _encode(self, slot, i, out_path, post, meta):
        slot.ev_out.synchronize()                       # the D2H copy of this slot has landed
        restored = slot.pin_out[i].numpy().copy()
        if post is not None:
            restored = post(slot.pin_in[i].numpy(), restored, meta)
        imwrite(restored, out_path, compress_level=self.png_level)

    # ---- driver --------------------------------------------------------------------------------------------------------------
    @torch.no_grad()
    def restore(self, paths, out_paths, w=0.5, adain=True, post=None, metas=None, strict=False):
        from . import ops
        assert len(paths) == len(out_paths)
        n, B = len(paths), self.batch
        nb = (n + B - 1) // B
        stats = {'faces': n, 'batches': nb, 'failures': 0, 'wait_decode_s': 0.0, 'wait_slot_s': 0.0}
        t0 = time.perf_counter()
        with ThreadPoolExecutor(self.workers) as pool:
            decodes = {}

            def submit_decode(k):
                slot = self.slots[k % len(self.slots)]
                t = time.perf_counter()
                for f in slot.writes:                   # the slot is free once its previous batch has been written out
                    f.result()
                stats['wait_slot_s'] += time.perf_counter() - t
                slot.writes = []
                lo, hi = k * B, min(n, (k + 1) * B)
                decodes[k] = [pool.submit(self._decode, paths[j], slot.pin_in[j - lo]) for j in range(lo, hi)]

            ahead = len(self.slots) - 1
            for k in range(min(ahead, nb)):
                submit_decode(k)
            for k in range(nb):
                slot = self.slots[k % len(self.slots)]
                lo, hi = k * B, min(n, (k + 1) * B)
                m = hi - lo
                t = time.perf_counter()
                for f in decodes.pop(k):
                    f.result()
                stats['wait_decode_s'] += time.perf_counter() - t
                with torch.cuda.stream(self.h2d):
                    slot.dev_in[:m].copy_(slot.pin_in[:m], non_blocking=True)
                    slot.ev_in.record(self.h2d)
                cur = torch.cuda.current_stream(self.device)
                cur.wait_event(slot.ev_in)
                x = ops.img_u8_to_tensor(slot.dev_in[:m])
                try:
                    out = self.net(x, w=w, adain=adain)[0]
                except Exception as error:  # noqa: BLE001 -- the reference prints the error and returns the input face
                    if strict:
                        raise
                    print(f'\tFailed inference for CodeFormer: {error}')
                    stats['failures'] += m
                    out = x
                slot.dev_out[:m].copy_(ops.tensor_to_img_u8(out))
                slot.ev_done.record(cur)
                with torch.cuda.stream(self.d2h):
                    self.d2h.wait_event(slot.ev_done)
                    slot.pin_out[:m].copy_(slot.dev_out[:m], non_blocking=True)
                    slot.ev_out.record(self.d2h)
                slot.writes = [pool.submit(self._encode, slot, j - lo, out_paths[j], post, None if metas is None else metas[j])
                               for j in range(lo, hi)]
                if k + ahead < nb:
                    submit_decode(k + ahead)
            for slot in self.slots:
                for f in slot.writes:
                    f.result()
                slot.writes = []
        torch.cuda.synchronize(self.device)
        stats['seconds'] = time.perf_counter() - t0
        stats['faces_per_s'] = n / stats['seconds'] if stats['seconds'] > 0 else float('inf')
        return stats
