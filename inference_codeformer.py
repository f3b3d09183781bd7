"""CodeFormer face restoration -- MI355X-native drop-in for the reference's inference_codeformer.py.

Same flags, same result tree (results/<input>_<w>/restored_faces/<basename>.png) for the `--has_aligned` path
(reference: inference_codeformer.py:55-274).  What is new underneath:
  * faces are restored in BATCHES (--batch_size, default 16 on GPU) instead of one forward per face;
  * on an MI355X the uint8<->tensor boundary (img2tensor+normalize, tensor2img) runs as HIP kernels on the device
    (cf_img_u8_to_tensor / cf_tensor_to_img_u8) and the network is codeformer_amd's HIP path;
  * under torchrun the face list is sharded over ranks (one process per GPU) -- each rank writes its own results.
Whole images (and directories of extracted video frames) take the reference's full flow: RetinaFace detection + alignment fit on the
host (facelib), crop warp / batched restoration / paste-back on the device; --affine_npz replaces the detector by a file of matrices.
Video container files (.mp4 ...) need ffmpeg, which this image lacks: extract the frames to a directory first.
"""
import argparse
import glob
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from basicsr.utils import imwrite  # noqa: E402
from basicsr.utils.download_util import load_file_from_url  # noqa: E402
from basicsr.utils.img_util import imread_bgr, resize_bilinear  # noqa: E402
from basicsr.utils.misc import get_device  # noqa: E402
from basicsr.utils.registry import ARCH_REGISTRY  # noqa: E402
from codeformer_amd.cli import faces_to_tensor, tensor_to_faces  # noqa: E402
from codeformer_amd.utils.face_misc import AlignedFaceHelper, is_gray  # noqa: E402

pretrain_model_url = {
    'restoration': 'https://github.com/sczhou/CodeFormer/releases/download/v0.1.0/codeformer.pth',
}
IMAGE_EXT = ('jpg', 'jpeg', 'png', 'JPG', 'JPEG', 'PNG')
VIDEO_EXT = ('mp4', 'mov', 'avi', 'MP4', 'MOV', 'AVI')


def parse_args(argv=None):
    p = argparse.ArgumentParser()
    p.add_argument('-i', '--input_path', type=str, default='./inputs/whole_imgs',
                   help='Input image, video or folder. Default: inputs/whole_imgs')
    p.add_argument('-o', '--output_path', type=str, default=None, help='Output folder. Default: results/<input_name>_<w>')
    p.add_argument('-w', '--fidelity_weight', type=float, default=0.5, help='Balance the quality and fidelity. Default: 0.5')
    p.add_argument('-s', '--upscale', type=int, default=2, help='The final upsampling scale of the image. Default: 2')
    p.add_argument('--has_aligned', action='store_true', help='Input are cropped and aligned faces. Default: False')
    p.add_argument('--only_center_face', action='store_true', help='Only restore the center face. Default: False')
    p.add_argument('--draw_box', action='store_true', help='Draw the bounding box for the detected faces. Default: False')
    p.add_argument('--detection_model', type=str, default='retinaface_resnet50')
    p.add_argument('--bg_upsampler', type=str, default='None', help='Background upsampler. Optional: realesrgan')
    p.add_argument('--face_upsample', action='store_true', help='Face upsampler after enhancement. Default: False')
    p.add_argument('--bg_tile', type=int, default=400, help='Tile size for background sampler. Default: 400')
    p.add_argument('--suffix', type=str, default=None, help='Suffix of the restored faces. Default: None')
    p.add_argument('--save_video_fps', type=float, default=None, help='Frame rate for saving video. Default: None')
    # --- additions (do not change the behaviour of the flags above) ---
    p.add_argument('--batch_size', type=int, default=None, help='Faces per forward. Default: 16 on GPU, 1 on CPU')
    p.add_argument('--device', type=str, default=None, help="Override the device pick (e.g. 'cpu')")
    p.add_argument('--random_init_seed', type=int, default=None,
                   help='Use torch.manual_seed(SEED) random weights when weights/CodeFormer/codeformer.pth is absent '
                        '(plumbing runs on boxes without the checkpoint)')
    p.add_argument('--affine_npz', type=str, default=None,
                   help='whole-image inputs: .npz mapping image basename -> (k,2,3) alignment matrices, instead of running the detector')
    p.add_argument('--det_device', type=str, default='auto',
                   help="where the face detector runs: 'auto' = the compute device, as the reference places it (stock torch ops through "
                        "MIOpen: 193 frames/s for RetinaFace-ResNet50 at 640x1138 on an MI355X), 'cpu' = the host cores (2 frames/s on 64 threads)")
    p.add_argument('--io_workers', type=int, default=None, help='PNG decode / encode worker threads of the GPU pipeline')
    p.add_argument('--strict', action='store_true', help='Raise on inference errors instead of returning the input face')
    p.add_argument('--frame_window', type=int, default=32, help='Whole-image / video path: images restored, pasted back and written per window '
                   '(bounds host memory; faces are batched 16 per forward across the images of a window)')
    return p.parse_args(argv)


def set_realesrgan(args, device, random_init_seed=None):
    """RealESRGANer(RRDBNet x2) as the reference builds it (inference_codeformer.py:19-52): half operands on a GPU, fp32 on CPU.
    Returns None (with a note) when the checkpoint is absent and no seeded random init was requested."""
    from basicsr.archs.rrdbnet_arch import RRDBNet
    from basicsr.utils.realesrgan_utils import RealESRGANer
    model = RRDBNet(num_in_ch=3, num_out_ch=3, num_feat=64, num_block=23, num_grow_ch=32, scale=2)
    url = 'https://github.com/sczhou/CodeFormer/releases/download/v0.1.0/RealESRGAN_x2plus.pth'
    try:
        return RealESRGANer(scale=2, model_path=url, model=model, tile=args.bg_tile, tile_pad=40, pre_pad=0,
                            half=device.type == 'cuda', device=device)
    except FileNotFoundError as e:
        if random_init_seed is None:
            print(f'NOTE: {e}; the Real-ESRGAN upsampler is not used on the --has_aligned path and is skipped')
            return None
        print('WARNING: RealESRGAN_x2plus.pth not found -- using random weights for the (unused) upsampler')
        return RealESRGANer(scale=2, model_path=None, model=model, tile=args.bg_tile, tile_pad=40, pre_pad=0,
                            half=device.type == 'cuda', device=device)


def build_detector(args, device):
    """The RetinaFace of facelib (init_detection_model reads weights/facelib/) -- host-side code on stock torch ops, placed on the
    compute device like the reference places it (--det_device cpu keeps it on the host cores); with --random_init_seed a missing
    checkpoint becomes seeded random weights (plumbing runs: such a detector finds nothing sensible)."""
    from facelib.detection import RetinaFace, init_detection_model
    det_device = device if args.det_device == 'auto' else torch.device(args.det_device)
    try:
        return init_detection_model(args.detection_model, half=False, device=det_device)
    except FileNotFoundError:
        if args.random_init_seed is None:
            raise
        print(f'WARNING: detector checkpoint not found -- using torch.manual_seed({args.random_init_seed}) random weights')
        torch.manual_seed(args.random_init_seed)
        return RetinaFace(network_name=args.detection_model.replace('retinaface_', ''), device=det_device)


def build_parser(args, device):
    """ParseNet for the paste-back masks (the reference's helper is built with use_parse=True, inference_codeformer.py:155-162).
    Without its checkpoint the square soft mask alone is used, and the run says so."""
    from facelib.parsing import ParseNet, init_parsing_model
    try:
        return init_parsing_model(model_name='parsenet', device=device)
    except FileNotFoundError as e:
        if args.random_init_seed is None:
            print(f'NOTE: {e}; the paste-back uses the square soft mask only')
            return None
        torch.manual_seed(args.random_init_seed)
        return ParseNet(in_size=512, out_size=512, parsing_ch=19).eval().to(device)


def restore_whole_images(args, input_img_list, result_root, w):
    """Whole images / extracted video frames (reference loop: inference_codeformer.py:165-262).  Faces are detected and aligned on the
    HOST (RetinaFace + the LMedS similarity fit, facelib) -- or their alignment matrices come from a file (--affine_npz: image
    basename without extension -> (k, 2, 3) frame -> 512-face matrices, FaceRestoreHelper.affine_matrices) -- and everything that
    touches pixels runs on the device: crops are cut, restored in 16-face batches ACROSS images and pasted back (codeformer_amd.video).
    Result tree as the reference's: cropped_faces/, restored_faces/ (<name>_<idx>.png) and final_results/<name>.png."""
    device = torch.device(args.device) if args.device else get_device()
    if device.type != 'cuda':
        raise NotImplementedError('the whole-image path runs on a ROCm device (alignment warp and paste-back are HIP kernels)')
    if args.draw_box or args.face_upsample:
        raise NotImplementedError('--draw_box / --face_upsample inside the paste-back are not built')
    from codeformer_amd.video import VideoRestorer, frame_shard
    from facelib.utils.face_restoration_helper import FaceRestoreHelper
    table = np.load(args.affine_npz) if args.affine_npz else None
    rank, world = int(os.environ.get('RANK', 0)), int(os.environ.get('WORLD_SIZE', 1))
    if world > 1:
        from codeformer_amd import parallel
        _, _, device = parallel.init_distributed(device=device)
    mine = [input_img_list[i] for i in frame_shard(len(input_img_list), rank, world)]       # frames are the unit of sharding
    net = build_net(device, args)
    parser = build_parser(args, device)
    bg = None
    if args.bg_upsampler == 'realesrgan':
        ups = set_realesrgan(args, device, args.random_init_seed)
        if ups is not None:
            def bg(frame):
                img = ups.enhance(frame, outscale=args.upscale)[0]
                return resize_bilinear(img, (frame.shape[1] * args.upscale, frame.shape[0] * args.upscale))
    # (the helper lives on the compute device: a detector placed there reads the uploaded frame through the INTER_AREA / INTER_LINEAR
    # kernels instead of a host-side resize; crops and paste-back are the VideoRestorer's)
    helper = FaceRestoreHelper(args.upscale, face_size=512, crop_ratio=(1, 1), det_model=args.detection_model, save_ext='png',
                               use_parse=False, device=device, face_detector=build_detector(args, device) if table is None else False)
    vr = VideoRestorer(net, device, upscale=args.upscale, batch_size=args.batch_size or 16, bg_upsampler=bg,
                       use_parse=parser is not None, face_parse=parser)
    totals = {'frames': 0, 'faces': 0, 'forward_calls': 0}
    failed = []

    def flush(frames, affs, names, grays):
        """Restore, paste and write one window of frames, then let go of it: host memory is bounded by the window, not by the
        length of the clip (the reference streams one image at a time; a window keeps the 16-face forwards full across images)."""
        try:
            outs = vr.restore(frames, affs, w=w, keep_faces=True, gray=grays)
            per_frame = vr.faces_out
        except Exception as error:   # the reference's per-image fallback (inference_codeformer.py:207-209): keep going, report at the end
            if args.strict:
                raise
            if len(names) > 1:       # a window failed: redo it image by image so that only the offending image is lost
                for one in zip(frames, affs, names, grays):
                    flush(*[[v] for v in one])
                return
            print(f'\tFailed inference for CodeFormer ({names[0]}): {error}')
            failed.extend(names)
            return
        for name, img, (crops, faces) in zip(names, outs, per_frame):
            for idx, (crop, face) in enumerate(zip(crops, faces)):
                imwrite(crop, os.path.join(result_root, 'cropped_faces', f'{name}_{idx:02d}.png'))
                face_name = f'{name}_{idx:02d}.png' if args.suffix is None else f'{name}_{idx:02d}_{args.suffix}.png'
                imwrite(face, os.path.join(result_root, 'restored_faces', face_name))
            out_name = name if args.suffix is None else f'{name}_{args.suffix}'
            imwrite(img, os.path.join(result_root, 'final_results', f'{out_name}.png'))
        for k in totals:
            totals[k] += vr.stats.get(k, 0)
        vr.faces_out = None

    window = max(1, int(getattr(args, 'frame_window', 32) or 32))
    frames, affs, names, grays = [], [], [], []
    for n_done, p in enumerate(mine, 1):
        name = os.path.splitext(os.path.basename(p))[0]
        helper.clean_all()
        helper.read_image(p)                                       # (short side below 512: enlarged, as the reference does)
        if table is None:
            helper.get_face_landmarks_5(only_center_face=args.only_center_face, resize=640, eye_dist_threshold=5)
            a = np.asarray(helper.estimate_affines(), dtype=np.float64).reshape(-1, 2, 3)
        else:
            a = np.asarray(table[name], dtype=np.float64).reshape(-1, 2, 3) if name in table.files else np.zeros((0, 2, 3))
            if args.only_center_face and a.shape[0] > 1:
                a = a[:1]
        frames.append(helper.input_img)
        grays.append(helper.is_gray)
        affs.append(a)
        names.append(name)
        print(f'[{n_done}/{len(mine)}] Processing: {os.path.basename(p)}\n\tdetect {a.shape[0]} faces')
        if len(frames) == window:
            flush(frames, affs, names, grays)
            frames, affs, names, grays = [], [], [], []
    if frames:
        flush(frames, affs, names, grays)
    print(f"{totals['faces']} faces of {totals['frames']} images in {totals['forward_calls']} forward calls")
    if failed:
        print(f'{len(failed)} image(s) failed: {failed[:8]}{" ..." if len(failed) > 8 else ""}')
    print(f'\nAll results are saved in {result_root}')
    return 0


def collect_inputs(args):
    w = args.fidelity_weight
    path = args.input_path
    if path.endswith(IMAGE_EXT):
        return [path], f'results/test_img_{w}'
    if path.endswith(VIDEO_EXT):
        raise NotImplementedError('video container input needs ffmpeg (absent on this image): extract the frames to a directory and '
                                  'pass the directory')
    path = path[:-1] if path.endswith('/') else path
    return sorted(glob.glob(os.path.join(path, '*.[jpJP][pnPN]*[gG]'))), f'results/{os.path.basename(path)}_{w}'


def build_net(device, args):
    net = ARCH_REGISTRY.get('CodeFormer')(dim_embd=512, codebook_size=1024, n_head=8, n_layers=9,
                                          connect_list=['32', '64', '128', '256'])
    try:
        ckpt_path = load_file_from_url(url=pretrain_model_url['restoration'], model_dir='weights/CodeFormer', progress=True,
                                       file_name=None)
        net.load_state_dict(torch.load(ckpt_path, map_location='cpu')['params_ema'])
    except FileNotFoundError:
        if args.random_init_seed is None:
            raise
        print(f'WARNING: codeformer.pth not found -- using torch.manual_seed({args.random_init_seed}) random weights')
        torch.manual_seed(args.random_init_seed)
        net = ARCH_REGISTRY.get('CodeFormer')(dim_embd=512, codebook_size=1024, n_head=8, n_layers=9,
                                              connect_list=['32', '64', '128', '256'])
    return net.to(device).eval()


def main(argv=None):
    args = parse_args(argv)
    device = torch.device(args.device) if args.device else get_device()
    w = args.fidelity_weight
    input_img_list, result_root = collect_inputs(args)
    if args.output_path is not None:
        result_root = args.output_path
    if len(input_img_list) == 0:
        raise FileNotFoundError('No input image/video is found...\n'
                                '\tNote that --input_path for video should end with .mp4|.mov|.avi')
    if not args.has_aligned:
        return restore_whole_images(args, input_img_list, result_root, w)
    # The reference builds the Real-ESRGAN upsampler for these flags (inference_codeformer.py:112-124) but only ever USES it in the
    # paste-back of whole images (:217-229): on the --has_aligned path it never runs.  Same here: build it when its checkpoint is
    # present (weights/realesrgan/RealESRGAN_x2plus.pth), say so when it is not.
    bg_upsampler = set_realesrgan(args, device, args.random_init_seed) if args.bg_upsampler == 'realesrgan' else None
    face_upsampler = (bg_upsampler or set_realesrgan(args, device, args.random_init_seed)) if args.face_upsample else None

    rank, world = int(os.environ.get('RANK', 0)), int(os.environ.get('WORLD_SIZE', 1))
    if world > 1:
        from codeformer_amd import parallel
        _, _, device = parallel.init_distributed(device=device)
        b = parallel.shard_bounds(len(input_img_list), world)
        my_list = input_img_list[b[rank]:b[rank + 1]]
        first = b[rank]
    else:
        my_list, first = input_img_list, 0

    net = build_net(device, args)
    print(f'Background upsampling: {bg_upsampler is not None}, Face upsampling: {args.face_upsample}')
    face_helper = AlignedFaceHelper()
    bs = args.batch_size or (16 if device.type == 'cuda' else 1)
    total = len(input_img_list)
    failures = 0

    def finish_face(face, out, g):
        """What the reference does with a restored crop before saving it (inference_codeformer.py:211-214 +
        face_restoration_helper.py:364-369): gray inputs get the gray colour transfer."""
        helper = AlignedFaceHelper()
        helper.is_gray = g
        helper.cropped_faces = [face]
        helper.add_restored_face(out.astype('uint8'), face)
        r = helper.restored_faces[0]
        return r if r.dtype == np.uint8 else np.clip(np.round(r), 0, 255).astype('uint8')

    def out_name(img_path):
        name = os.path.splitext(os.path.basename(img_path))[0]
        return os.path.join(result_root, 'restored_faces', f'{name}.png' if args.suffix is None else f'{name}_{args.suffix}.png')

    if device.type == 'cuda':
        # overlapped pipeline: PNG decode / encode in a worker pool, uint8 over PCIe through pinned staging on side streams
        from codeformer_amd.pipeline import AlignedFacePipeline
        for j, img_path in enumerate(my_list):
            print(f'[{first + j + 1}/{total}] Processing: {os.path.basename(img_path)}')
        pipe = AlignedFacePipeline(net, device, batch_size=bs, workers=args.io_workers)

        def post(face, out, meta):
            g = is_gray(face, threshold=10)
            if g:
                print('Grayscale input: True')
            return finish_face(face, out, g)

        st = pipe.restore(my_list, [out_name(q) for q in my_list], w=w, adain=True, post=post, strict=args.strict)
        failures = st['failures']
        print(f"{st['faces']} faces in {st['seconds']:.2f} s = {st['faces_per_s']:.1f} faces/s including PNG decode / encode "
              f"(host waited {st['wait_decode_s']:.2f} s for decodes, {st['wait_slot_s']:.2f} s for writes)")
        my_list = []

    for s in range(0, len(my_list), bs):
        chunk = my_list[s:s + bs]
        faces, grays = [], []
        for j, img_path in enumerate(chunk):
            img_name = os.path.basename(img_path)
            print(f'[{first + s + j + 1}/{total}] Processing: {img_name}')
            img = resize_bilinear(imread_bgr(img_path), (512, 512))
            g = is_gray(img, threshold=10)
            if g:
                print('Grayscale input: True')
            faces.append(img)
            grays.append(g)
        x = faces_to_tensor(faces, device)
        try:
            with torch.no_grad():
                restored = tensor_to_faces(net(x, w=w, adain=True)[0])
        except Exception as error:  # the reference swallows the error and returns the input face (F7)
            if args.strict:
                raise
            print(f'\tFailed inference for CodeFormer: {error}')
            failures += len(faces)
            restored = tensor_to_faces(x)
        for img_path, face, out, g in zip(chunk, faces, restored, grays):
            imwrite(finish_face(face, out, g), out_name(img_path))
    if failures:
        print(f'WARNING: {failures} face(s) fell back to the unrestored input')
    print(f'\nAll results are saved in {result_root}')
    return failures


if __name__ == '__main__':
    sys.exit(1 if main() else 0)
