"""FaceRestoreHelper -- the whole-image pipeline object of the reference (facelib/utils/face_restoration_helper.py:54-525) with the
work split the north star prescribes: DETECTION and the alignment fit on the host, everything that touches pixels on the device.

    read_image -> get_face_landmarks_5 (RetinaFace on the host) -> align_warp_face (LMedS similarity on the host, all crops of the
    frame cut by ONE warp launch) -> [CodeFormer on the crops] -> add_restored_face(s) -> get_inverse_affine ->
    paste_faces_to_input_image (warp / erode / blur / blend kernels over each face's bounding box)

Method names, arguments and the lists a caller reads (`all_landmarks_5`, `det_faces`, `affine_matrices`, `inverse_affine_matrices`,
`cropped_faces`, `restored_faces`) follow the reference; `cropped_faces` / `restored_faces` hold uint8 HWC BGR arrays on the host
when asked for (`cropped_faces` property) and stay device tensors otherwise.

Not built (raise NotImplementedError): `det_model='dlib'`, `pad_blur=True` (cv2.boxFilter padding of blurry inputs), `draw_box`,
`face_upsampler` inside the paste, 16-bit / RGBA inputs.
"""
import os

import numpy as np
import torch

from ...utils.face_misc import adain_npy, bgr2gray, is_gray
from ...utils.img_util import imread_bgr, imwrite, resize_area, resize_bilinear
from ..align import estimate_affine_partial_2d
from ..detection import init_detection_model
from ..paste import DeviceFaceHelper

# standard 5 landmarks of FFHQ faces at 512 x 512 (facexlib; :81-85) and the 3-point variant (:79)
_TEMPLATE_5 = ((192.98138, 239.94708), (318.90277, 240.1936), (256.63416, 314.01935), (201.26117, 371.41043), (313.08905, 371.15118))
_TEMPLATE_3 = ((192, 240), (319, 240), (257, 371))


def _clip(v, hi):
    return 0 if v < 0 else (hi if v > hi else v)


def get_largest_face(det_faces, h, w):
    """(face, index) of the detection with the largest box area after clipping to the image (:19-37)."""
    areas = [(_clip(d[2], w) - _clip(d[0], w)) * (_clip(d[3], h) - _clip(d[1], h)) for d in det_faces]
    k = areas.index(max(areas))
    return det_faces[k], k


def get_center_face(det_faces, h=0, w=0, center=None):
    """(face, index) of the detection whose box centre is nearest to `center` (default: the image centre) (:40-51)."""
    c = np.array(center) if center is not None else np.array([w / 2, h / 2])
    dist = [np.linalg.norm(np.array([(d[0] + d[2]) / 2, (d[1] + d[3]) / 2]) - c) for d in det_faces]
    k = dist.index(min(dist))
    return det_faces[k], k


class FaceRestoreHelper(object):

    def __init__(self, upscale_factor, face_size=512, crop_ratio=(1, 1), det_model='retinaface_resnet50', save_ext='png',
                 template_3points=False, pad_blur=False, use_parse=False, device=None, face_detector=None, face_parse=None,
                 det_device='cpu'):
        """Extra arguments: `face_detector` / `face_parse` take ready models (otherwise init_detection_model / init_parsing_model read
        weights/facelib/; face_detector=False builds none); `det_device` is where the detector runs (host by default)."""
        if det_model == 'dlib':
            raise NotImplementedError('det_model dlib is not built')
        if pad_blur:
            raise NotImplementedError('pad_blur is not built')
        if tuple(crop_ratio) != (1, 1):
            raise NotImplementedError('crop_ratio other than (1, 1) is not built (the device crops are square)')
        self.template_3points, self.upscale_factor, self.crop_ratio = template_3points, int(upscale_factor), tuple(crop_ratio)
        self.face_size = (int(face_size), int(face_size))
        self.det_model, self.save_ext, self.pad_blur, self.use_parse = det_model, save_ext, False, use_parse
        self.face_template = np.array(_TEMPLATE_3 if template_3points else _TEMPLATE_5, dtype=np.float64) * (face_size / 512.0)
        if device is None:
            device = torch.device('cuda' if torch.cuda.is_available() else 'cpu')
        self.device = torch.device(device)
        if face_detector is None:
            face_detector = init_detection_model(det_model, half=False, device=det_device)
        self.face_detector = face_detector or None            # face_detector=False: no detector (alignment matrices come from outside)
        if face_parse is None and use_parse:
            from ..parsing import init_parsing_model
            face_parse = init_parsing_model(model_name='parsenet', device=self.device)
        self.face_parse = face_parse
        self._dev = None
        self.is_gray = False
        self.clean_all()

    def _device_helper(self):
        if self.device.type != 'cuda':
            raise RuntimeError('crop warp and paste-back are HIP kernels: FaceRestoreHelper needs a ROCm device for these steps')
        if self._dev is None:
            self._dev = DeviceFaceHelper(self.upscale_factor, self.face_size[0], self.device, self.use_parse, self.face_parse)
        return self._dev

    def set_upscale_factor(self, upscale_factor):
        self.upscale_factor = int(upscale_factor)
        self._dev = None

    def clean_all(self):
        self.all_landmarks_5, self.det_faces, self.affine_matrices, self.inverse_affine_matrices = [], [], [], []
        self.restored_faces, self.pad_input_imgs = [], []
        self._crops = None
        if self._dev is not None:
            self._dev.clean_all()

    # ---- read_image (:133-155) ------------------------------------------------------------------------------------------------
    def read_image(self, img):
        """img: path or uint8 HWC BGR array.  Gray / BGRA inputs are brought to 3 channels; images whose short side is below 512
        are enlarged to 512 (INTER_LINEAR) as the reference does."""
        if isinstance(img, str):
            img = imread_bgr(img)
        img = np.asarray(img)
        if img.dtype != np.uint8:
            raise NotImplementedError('16-bit inputs are not built')
        if img.ndim == 2:
            img = np.repeat(img[:, :, None], 3, axis=2)
        elif img.shape[2] == 4:
            img = img[:, :, 0:3]
        img = np.ascontiguousarray(img)
        self.is_gray = is_gray(img, threshold=10)
        if self.is_gray:
            print('Grayscale input: True')
        h, w = img.shape[:2]
        if min(h, w) < 512:
            f = 512.0 / min(h, w)
            # cv2.resize(img, (0, 0), fx=f, fy=f, INTER_LINEAR) (face_restoration_helper.py:159-161): dsize = round(f * size), coordinates
            # mapped with 1 / f on both axes
            img = resize_bilinear(img, (int(round(w * f)), int(round(h * f))), inv_scale=(f, f))
        self.input_img = img
        if self.device.type == 'cuda':
            self._device_helper().read_image(img)

    # ---- detection (:195-247) -------------------------------------------------------------------------------------------------
    def get_face_landmarks_5(self, only_keep_largest=False, only_center_face=False, resize=None, blur_ratio=0.01,
                             eye_dist_threshold=None):
        # When the detector lives on the compute device, so does its input: the frame was uploaded by read_image, the reduction to the
        # detector's working size is a kernel (cf_resize_area_u8 / cf_resize_linear_u8), and nothing crosses PCIe but the boxes.
        on_dev = self.device.type == 'cuda' and getattr(self.face_detector, 'device', torch.device('cpu')).type == 'cuda'
        if resize is None:
            scale, det_in = 1, (self._device_helper().input_img if on_dev else self.input_img)
        else:
            h, w = self.input_img.shape[0:2]
            scale = resize / min(h, w)
            size = (int(w * scale), int(h * scale))
            if on_dev:
                from ... import ops
                frame = self._device_helper().input_img
                det_in = ops.resize_area_u8(frame, size[1], size[0]) if scale < 1 else \
                    ops.f32_to_u8_trunc(ops.resize_linear_u8(frame, size[1], size[0]))
            else:
                det_in = resize_area(self.input_img, size) if scale < 1 else resize_bilinear(self.input_img, size)
        if self.face_detector is None:
            raise RuntimeError('this helper was built without a detector (face_detector=False)')
        with torch.no_grad():
            bboxes = self.face_detector.detect_faces(det_in)
        if bboxes is None or bboxes.shape[0] == 0:
            return 0
        bboxes = bboxes / scale
        npts = 3 if self.template_3points else 5
        for bbox in bboxes:
            eye_dist = np.linalg.norm([bbox[6] - bbox[8], bbox[7] - bbox[9]])   # the reference's indices as written (:223): (ly0 - ly1, lx1 - lx2)
            if eye_dist_threshold is not None and eye_dist < eye_dist_threshold:
                continue
            self.all_landmarks_5.append(np.array([[bbox[5 + 2 * j], bbox[6 + 2 * j]] for j in range(npts)]))
            self.det_faces.append(bbox[0:5])
        if len(self.det_faces) == 0:
            return 0
        h, w, _ = self.input_img.shape
        if only_keep_largest:
            self.det_faces, k = get_largest_face(self.det_faces, h, w)
            self.all_landmarks_5 = [self.all_landmarks_5[k]]
        elif only_center_face:
            self.det_faces, k = get_center_face(self.det_faces, h, w)
            self.all_landmarks_5 = [self.all_landmarks_5[k]]
        return len(self.all_landmarks_5)

    # ---- alignment (:320-362) -------------------------------------------------------------------------------------------------
    def estimate_affines(self):
        """The frame -> face similarity of every landmark set (cv2.estimateAffinePartial2D(..., method=cv2.LMEDS)[0], :329)."""
        self.affine_matrices = [estimate_affine_partial_2d(lm, self.face_template)[0] for lm in self.all_landmarks_5]
        return self.affine_matrices

    def align_warp_face(self, save_cropped_path=None, border_mode='constant'):
        if border_mode != 'constant':
            raise NotImplementedError("border_mode: only 'constant' (the gray border) is built")
        self.estimate_affines()
        self._crops = self._device_helper().align_warp_face(np.asarray(self.affine_matrices, dtype=np.float64).reshape(-1, 2, 3))
        if save_cropped_path is not None:
            path = os.path.splitext(save_cropped_path)[0]
            for idx, face in enumerate(self.cropped_faces):
                imwrite(face, f'{path}_{idx:02d}.{self.save_ext}')

    @property
    def cropped_faces_device(self):
        """uint8 (n, 512, 512, 3) BGR crops on the device (what the batched restoration consumes)."""
        return self._crops

    @property
    def cropped_faces(self):
        return [] if self._crops is None else list(self._crops.cpu().numpy())

    def get_inverse_affine(self, save_inverse_affine_path=None):
        self.inverse_affine_matrices = self._device_helper().get_inverse_affine()
        if save_inverse_affine_path is not None:
            path, _ = os.path.splitext(save_inverse_affine_path)
            for idx, m in enumerate(self.inverse_affine_matrices):
                torch.save(m, f'{path}_{idx:02d}.pth')

    # ---- restored faces (:364-369) ----------------------------------------------------------------------------------------------
    def add_restored_face(self, restored_face, input_face=None):
        """One uint8 HWC BGR face (host array, as the reference's loop hands it over)."""
        if self.is_gray:
            restored_face = bgr2gray(restored_face)
            if input_face is not None:
                restored_face = adain_npy(restored_face, input_face)
        self.restored_faces.append(restored_face)

    def add_restored_faces(self, restored):
        """All faces of the frame at once as a uint8 (n, 512, 512, 3) device tensor (cf_tensor_to_img_u8 output); gray frames take the
        reference's gray colour transfer on the host."""
        if self.is_gray:
            crops = self.cropped_faces
            for k, face in enumerate(restored.cpu().numpy()):
                self.add_restored_face(face, crops[k])
        else:
            self.restored_faces = restored

    # ---- paste-back (:372-499) ------------------------------------------------------------------------------------------------------
    def paste_faces_to_input_image(self, save_path=None, upsample_img=None, draw_box=False, face_upsampler=None):
        if draw_box or face_upsampler is not None:
            raise NotImplementedError('draw_box / face_upsampler inside the paste-back are not built')
        dev = self._device_helper()
        faces = self.restored_faces
        if not torch.is_tensor(faces):
            arr = [np.clip(np.round(np.asarray(f, dtype=np.float32)), 0, 255).astype(np.uint8) if np.asarray(f).dtype != np.uint8
                   else np.asarray(f) for f in faces]
            faces = torch.from_numpy(np.stack(arr)).to(self.device) if arr else \
                torch.empty(0, self.face_size[1], self.face_size[0], 3, dtype=torch.uint8, device=self.device)
        dev.add_restored_faces(faces.contiguous())
        if not dev.inverse_affine_matrices:
            self.get_inverse_affine()
        if upsample_img is not None:
            h, w = self.input_img.shape[:2]
            upsample_img = resize_bilinear(np.ascontiguousarray(upsample_img), (w * self.upscale_factor, h * self.upscale_factor))
        out = dev.paste_faces_to_input_image(upsample_img=upsample_img)
        if save_path is not None:
            imwrite(out, f'{os.path.splitext(save_path)[0]}.{self.save_ext}')
        return out
