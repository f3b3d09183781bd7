"""RetinaFace detector: network + single-image and batched detection (reference: facelib/detection/retinaface/retinaface.py:20-372).

Same constructor, configs, `state_dict` keys (body.* / fpn.* / ssh1-3.* / ClassHead|BboxHead|LandmarkHead.N.conv1x1.*) and result
conventions as the reference: `detect_faces(bgr image)` -> (k, 15) rows [x1, y1, x2, y2, score, 5 x (lx, ly)] in pixels of the given
image, score-sorted and NMS-filtered; `batched_detect_faces(frames)` -> per-frame lists.  Detection is the host-code half of the
pipeline (north star: "facelib detection/alignment left on host"): the module is stock torch ops and runs on whatever device it is
placed on -- the entrypoint puts it on the compute device, as the reference does (MIOpen: 193 frames/s at 640x1138 on an MI355X,
2 frames/s on 64 host threads; tools/detector_bench.py) -- and takes device-resident frames; the crops its boxes yield are cut,
restored and pasted back by the HIP path (codeformer_amd.facelib.paste).

Not built: `align_multi` (112x112 crops through matlab_cp2tform; unused by the restoration entrypoints).
"""
import numpy as np
import torch
from torch import nn
from torch.nn import functional as F

from ....utils.img_util import resize_linear_f32
from .retinaface_net import (FPN, SSH, MobileNetTrunk, ResNet50Trunk, make_bbox_head, make_class_head, make_landmark_head)
from .retinaface_utils import PriorBox, batched_decode, batched_decode_landm, decode, decode_landm, py_cpu_nms

_COMMON = dict(min_sizes=[[16, 32], [64, 128], [256, 512]], steps=[8, 16, 32], variance=[0.1, 0.2], clip=False, loc_weight=2.0,
               gpu_train=True)


def generate_config(network_name):
    """cfg_mnet / cfg_re50 of retinaface.py:20-74 (the training fields are carried along unchanged)."""
    if network_name == 'mobile0.25':
        return dict(_COMMON, name='mobilenet0.25', batch_size=32, ngpu=1, epoch=250, decay1=190, decay2=220, image_size=640,
                    return_layers={'stage1': 1, 'stage2': 2, 'stage3': 3}, in_channel=32, out_channel=64)
    if network_name == 'resnet50':
        return dict(_COMMON, name='Resnet50', batch_size=24, ngpu=4, epoch=100, decay1=70, decay2=90, image_size=840,
                    return_layers={'layer2': 1, 'layer3': 2, 'layer4': 3}, in_channel=256, out_channel=256)
    raise NotImplementedError(f'network_name={network_name}')


class RetinaFace(nn.Module):

    def __init__(self, network_name='resnet50', half=False, phase='test', device='cpu'):
        super().__init__()
        self.half_inference = half
        cfg = generate_config(network_name)
        self.backbone = cfg['name']
        self.model_name = f'retinaface_{network_name}'
        self.cfg, self.phase = cfg, phase
        self.target_size, self.max_size = 1600, 2150
        self.resize, self.scale, self.scale1 = 1., None, None
        self.body = MobileNetTrunk() if cfg['name'] == 'mobilenet0.25' else ResNet50Trunk()
        c = cfg['in_channel']
        self.fpn = FPN([c * 2, c * 4, c * 8], cfg['out_channel'])
        for i in (1, 2, 3):
            setattr(self, f'ssh{i}', SSH(cfg['out_channel'], cfg['out_channel']))
        self.ClassHead = make_class_head(fpn_num=3, inchannels=cfg['out_channel'])
        self.BboxHead = make_bbox_head(fpn_num=3, inchannels=cfg['out_channel'])
        self.LandmarkHead = make_landmark_head(fpn_num=3, inchannels=cfg['out_channel'])
        self.register_buffer('mean_tensor', torch.tensor([[[[104.]], [[117.]], [[123.]]]]), persistent=False)
        self.to(device)
        self.eval()
        if half:
            self.half()

    @property
    def device(self):
        return self.mean_tensor.device

    def forward(self, inputs):
        """(B,3,H,W) mean-subtracted BGR -> (bbox (B,n,4), class probabilities (B,n,2), landmarks (B,n,10)); n = anchors of the three
        pyramid levels in level-major, row-major, anchor-minor order (the order PriorBox emits)."""
        pyramid = self.fpn(self.body(inputs))
        feats = [self.ssh1(pyramid[0]), self.ssh2(pyramid[1]), self.ssh3(pyramid[2])]
        bbox = torch.cat([h(f) for h, f in zip(self.BboxHead, feats)], dim=1)
        cls = torch.cat([h(f) for h, f in zip(self.ClassHead, feats)], dim=1)
        ldm = torch.cat([h(f) for h, f in zip(self.LandmarkHead, feats)], dim=1)
        return (bbox, cls, ldm) if self.phase == 'train' else (bbox, F.softmax(cls, dim=-1), ldm)

    # ---- shared -----------------------------------------------------------------------------------------------------------------
    def _run(self, inputs):
        """Network + anchors for a mean-subtracted batch; also sets the pixel scales (retinaface.py:122-139)."""
        h, w = inputs.shape[2:]
        dev = self.device
        self.scale = torch.tensor([w, h] * 2, dtype=torch.float32, device=dev)
        self.scale1 = torch.tensor([w, h] * 5, dtype=torch.float32, device=dev)
        inputs = inputs.to(dev)
        if self.half_inference:
            inputs = inputs.half()
        with torch.no_grad():
            loc, conf, landmarks = self(inputs)
        priors = PriorBox(self.cfg, image_size=inputs.shape[2:]).forward().to(dev)
        return loc, conf, landmarks, priors

    def _test_scale(self, h, w, use_origin_size):
        """The detector's own rescale rule (:150-157): short side to target_size unless the long side would exceed max_size."""
        smin, smax = min(h, w), max(h, w)
        r = float(self.target_size) / float(smin)
        if np.round(r * smax) > self.max_size:
            r = float(self.max_size) / float(smax)
        return 1 if use_origin_size else r

    # ---- single image (:142-213) ------------------------------------------------------------------------------------------------
    def transform(self, image, use_origin_size):
        if torch.is_tensor(image):                                 # uint8 / float (H, W, 3) BGR tensor, e.g. a frame that already lives on the device
            resize = self._test_scale(image.shape[0], image.shape[1], use_origin_size)
            if resize != 1:
                raise NotImplementedError('tensor inputs are taken at their own size (use_origin_size=True)')
            return image.permute(2, 0, 1).unsqueeze(0).float(), resize
        if not isinstance(image, np.ndarray):                      # PIL image: RGB -> BGR
            image = np.asarray(image)[:, :, ::-1]
        image = image.astype(np.float32)
        resize = self._test_scale(image.shape[0], image.shape[1], use_origin_size)
        if resize != 1:
            size = (int(round(image.shape[1] * resize)), int(round(image.shape[0] * resize)))
            image = resize_linear_f32(image, size, inv_scale=(resize, resize))
        return torch.from_numpy(np.ascontiguousarray(image.transpose(2, 0, 1))).unsqueeze(0), resize

    def detect_faces(self, image, conf_threshold=0.8, nms_threshold=0.4, use_origin_size=True):
        """image: uint8 / float HWC BGR array, a PIL image, or an (H, W, 3) BGR tensor on any device.  Returns float32 (k, 15)."""
        image, self.resize = self.transform(image, use_origin_size)
        image = image.to(self.device)
        if self.half_inference:
            image = image.half()
        image = image - self.mean_tensor.to(image.dtype)
        loc, conf, landmarks, priors = self._run(image)
        boxes = (decode(loc.squeeze(0), priors, self.cfg['variance']) * self.scale / self.resize).cpu().numpy()
        scores = conf.squeeze(0).cpu().numpy()[:, 1]
        landmarks = (decode_landm(landmarks.squeeze(0), priors, self.cfg['variance']) * self.scale1 / self.resize).cpu().numpy()
        inds = np.where(scores > conf_threshold)[0]
        boxes, landmarks, scores = boxes[inds], landmarks[inds], scores[inds]
        order = scores.argsort()[::-1]
        boxes, landmarks, scores = boxes[order], landmarks[order], scores[order]
        dets = np.hstack((boxes, scores[:, np.newaxis])).astype(np.float32, copy=False)
        keep = py_cpu_nms(dets, nms_threshold)
        return np.concatenate((dets[keep, :], landmarks[keep]), axis=1)

    # ---- batched (:238-372) -----------------------------------------------------------------------------------------------------
    def batched_transform(self, frames, use_origin_size):
        """frames: list of PIL images, or a float32 tensor (n, h, w, c) in BGR.  Returns ((n, c, h', w') float32, resize)."""
        from_pil = not torch.is_tensor(frames) and not isinstance(frames[0], np.ndarray)
        if from_pil:
            frames = np.asarray([np.asarray(f)[:, :, ::-1] for f in frames], dtype=np.float32)
        elif not torch.is_tensor(frames):
            frames = np.asarray(frames, dtype=np.float32)
        h, w = frames[0].shape[0:2]
        resize = self._test_scale(h, w, use_origin_size)
        if torch.is_tensor(frames):
            t = frames.permute(0, 3, 1, 2).contiguous().float()
            if resize != 1:
                t = F.interpolate(t, scale_factor=resize)
            return t, resize
        if resize != 1:
            size = (int(round(w * resize)), int(round(h * resize)))
            frames = np.stack([resize_linear_f32(f, size, inv_scale=(resize, resize)) for f in frames])
        return torch.from_numpy(np.ascontiguousarray(frames.transpose((0, 3, 1, 2)))), resize

    def batched_detect_faces(self, frames, conf_threshold=0.8, nms_threshold=0.4, use_origin_size=True):
        """Returns (list of (k_i, 5) float32 [box, score], list of (k_i, 10) float32 landmarks); frames without a detection give empty
        arrays.  As in the reference the per-frame rows are NOT score-sorted before NMS's own ordering."""
        frames, self.resize = self.batched_transform(frames, use_origin_size)
        frames = frames.to(self.device)
        frames = frames - self.mean_tensor
        b_loc, b_conf, b_landmarks, priors = self._run(frames)
        priors = priors.unsqueeze(0)
        b_loc = batched_decode(b_loc, priors, self.cfg['variance']) * self.scale / self.resize
        b_landmarks = batched_decode_landm(b_landmarks, priors, self.cfg['variance']) * self.scale1 / self.resize
        b_conf = b_conf[:, :, 1]
        b_keep = b_conf > conf_threshold
        b_pred = torch.cat((b_loc, b_conf.unsqueeze(-1)), dim=2).float()
        final_boxes, final_landmarks = [], []
        for pred, landm, inds in zip(b_pred, b_landmarks, b_keep):
            pred, landm = pred[inds, :], landm[inds, :]
            if pred.shape[0] == 0:
                final_boxes.append(np.array([], dtype=np.float32))
                final_landmarks.append(np.array([], dtype=np.float32))
                continue
            boxes, landm = pred.cpu().numpy(), landm.float().cpu().numpy()
            keep = py_cpu_nms(boxes, nms_threshold)
            final_boxes.append(boxes[keep, :])
            final_landmarks.append(landm[keep])
        return final_boxes, final_landmarks
