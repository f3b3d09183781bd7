"""Whole-frame / video fan-out (BASELINE config 4; reference loop: inference_codeformer.py:165-229).

The reference walks the frames one by one and, inside a frame, restores one face per forward (:197-214).  Here the faces of MANY
frames are batched into full 16-face calls:

    frames (host uint8) --H2D--> device frame --align_warp_face (one launch per frame)--> crops --+
                                                                                                 | queue of (frame, face) crops
    CodeFormer.forward on 16 crops at a time, whatever frames they come from  <-------------------+
    restored crops regrouped per frame --> paste_faces_to_input_image on the device --> output frame

Detection / landmark estimation stays on the host (north star): `affines[i]` holds the (k_i, 2, 3) alignment matrices the host
detector produced for frame i.  Across GPUs the unit of sharding is the FRAME (contiguous blocks, `frame_shard`): frames are
independent, a frame's faces never leave the GPU that holds the frame, and no collective is needed until -- optionally -- the
finished frames are gathered.
"""
import numpy as np
import torch

from . import ops
from .facelib.paste import DeviceFaceHelper
from .parallel import shard_bounds


def frame_shard(n_frames, rank, world):
    """Contiguous block of frame indices owned by `rank` (the first ranks take the remainder)."""
    b = shard_bounds(n_frames, world)
    return range(b[rank], b[rank + 1])


class VideoRestorer:
    """restore(frames, affines) -> list of pasted uint8 frames (numpy, in input order).

    net: CodeFormer on the device; upscale: the reference's -s/--upscale (background resized with INTER_LINEAR unless `bg_upsampler`
    is given: callable(frame uint8 numpy) -> uint8 (h*upscale, w*upscale, 3)); use_parse: ParseNet-refined masks (needs face_parse)."""

    def __init__(self, net, device='cuda', upscale=2, batch_size=16, use_parse=False, face_parse=None, bg_upsampler=None):
        self.net, self.device, self.upscale, self.batch = net, torch.device(device), int(upscale), int(batch_size)
        self.use_parse, self.face_parse, self.bg_upsampler = use_parse, face_parse, bg_upsampler
        self.stats = {}

    def _helper(self):
        return DeviceFaceHelper(self.upscale, 512, self.device, self.use_parse, self.face_parse)

    @torch.no_grad()
    def restore(self, frames, affines, w=0.5, adain=True, return_tensors=False, keep_faces=False, gray=None):
        """keep_faces: also keep every frame's (crops, restored faces) as uint8 host arrays in `self.faces_out[i]` (the reference saves
        them next to the pasted image, inference_codeformer.py:232-247).  gray: per-frame flags; the faces of a gray frame take the
        reference's gray colour transfer (face_restoration_helper.py:364-369: bgr2gray + adain_npy against the crop) on the host and are
        rounded back to uint8 before the paste (the reference pastes the float result: at most half a grey level of difference)."""
        n = len(frames)
        assert len(affines) == n
        helpers, out = [None] * n, [None] * n
        self.faces_out = [None] * n
        pending = [0] * n                    # faces of frame i still in flight
        done = [None] * n                    # per frame: list of restored crops (views), by face index
        queue = []                           # (frame, face, crop view)
        calls = faces = 0

        def finish(i):
            h = helpers[i]
            k = len(done[i])
            faces_i = torch.stack(done[i]) if k else torch.empty(0, 512, 512, 3, dtype=torch.uint8, device=self.device)
            if k and gray is not None and gray[i]:
                from .utils.face_misc import adain_npy, bgr2gray
                crops_np = h.cropped_faces.cpu().numpy()
                moved = [adain_npy(bgr2gray(f), c) for f, c in zip(faces_i.cpu().numpy(), crops_np)]
                faces_i = torch.from_numpy(np.clip(np.rint(np.stack(moved)), 0, 255).astype(np.uint8)).to(self.device)
            if keep_faces:
                self.faces_out[i] = (h.cropped_faces.cpu().numpy(), faces_i.cpu().numpy())
            h.add_restored_faces(faces_i)
            bg = self.bg_upsampler(frames[i]) if self.bg_upsampler is not None else None
            out[i] = h.paste_faces_to_input_image(upsample_img=bg, return_tensor=return_tensors)
            helpers[i] = done[i] = None      # release the frame's device buffers

        def run(items):
            nonlocal calls, faces
            x = ops.img_u8_to_tensor(torch.stack([c for _, _, c in items]))
            y = ops.tensor_to_img_u8(self.net(x, w=w, adain=adain)[0])
            calls += 1
            faces += len(items)
            for j, (i, f, _) in enumerate(items):
                done[i][f] = y[j]
                pending[i] -= 1
                if pending[i] == 0:
                    finish(i)

        for i in range(n):
            h = helpers[i] = self._helper()
            h.read_image(frames[i])
            crops = h.align_warp_face(affines[i])
            k = crops.shape[0]
            pending[i], done[i] = k, [None] * k
            if k == 0:
                finish(i)
            queue.extend((i, f, crops[f]) for f in range(k))
            while len(queue) >= self.batch:
                run(queue[:self.batch])
                queue = queue[self.batch:]
        if queue:
            run(queue)                       # the only partial call of the clip
        self.stats = {'frames': n, 'faces': faces, 'forward_calls': calls}
        return out


def gather_frames(local_frames, n_frames, dst=0):
    """Optional last step under torch.distributed: the pasted frames (uint8 CUDA tensors of ONE size) of every rank's block to `dst`
    with the single gather of codeformer_amd.parallel (frames play the role of faces)."""
    from .parallel import gather_faces
    stacked = torch.stack(local_frames) if local_frames else None
    if stacked is None:
        raise ValueError('gather_frames: a rank without frames must pass an empty (0,H,W,3) tensor list is not supported; give every '
                         'rank at least one frame')
    return gather_faces(stacked, n_frames, dst=dst)
