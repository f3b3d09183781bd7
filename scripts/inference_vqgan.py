"""VQGAN (stage I) reconstruction of aligned 512x512 faces -- drop-in for the reference's scripts/inference_vqgan.py:12-59, the caller
of `VectorQuantizer.forward`: image -> Encoder -> nearest-code L2 quantisation -> Generator -> image.

Same flags and result files (<save_root>/<name>.png).  On an MI355X the faces go through the network in batches (--batch_size) with the
uint8 <-> tensor boundary on the device (cf_img_u8_to_tensor / cf_tensor_to_img_u8); on a CPU the stock-torch host path runs one face at
a time like the reference.  --random_init_seed replaces a missing checkpoint by seeded random weights (plumbing runs).
"""
import argparse
import glob
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from basicsr.utils import imwrite  # noqa: E402
from basicsr.utils.img_util import imread_bgr  # noqa: E402
from basicsr.utils.registry import ARCH_REGISTRY  # noqa: E402
from codeformer_amd.cli import faces_to_tensor, tensor_to_faces  # noqa: E402


def main(argv=None):
    p = argparse.ArgumentParser()
    p.add_argument('-i', '--test_path', type=str, default='datasets/ffhq/ffhq_512')
    p.add_argument('-o', '--save_root', type=str, default='./results/vqgan_rec')
    p.add_argument('--codebook_size', type=int, default=1024)
    p.add_argument('--ckpt_path', type=str, default='./experiments/pretrained_models/vqgan/net_g.pth')
    p.add_argument('--batch_size', type=int, default=None, help='faces per forward (default 16 on a GPU, 1 on the CPU)')
    p.add_argument('--device', type=str, default=None)
    p.add_argument('--random_init_seed', type=int, default=None)
    args = p.parse_args(argv)
    save_root = args.save_root.rstrip('/') or '/'
    os.makedirs(os.path.abspath(save_root), exist_ok=True)
    device = torch.device(args.device) if args.device else torch.device('cuda' if torch.cuda.is_available() else 'cpu')

    def build():
        return ARCH_REGISTRY.get('VQAutoEncoder')(512, 64, [1, 2, 2, 4, 4, 8], 'nearest', codebook_size=args.codebook_size)

    if os.path.exists(args.ckpt_path):
        vqgan = build()
        vqgan.load_state_dict(torch.load(args.ckpt_path, map_location='cpu')['params_ema'])
    elif args.random_init_seed is not None:
        print(f'WARNING: {args.ckpt_path} not found -- using torch.manual_seed({args.random_init_seed}) random weights')
        torch.manual_seed(args.random_init_seed)
        vqgan = build()
    else:
        raise FileNotFoundError(args.ckpt_path)
    vqgan = vqgan.to(device).eval()

    paths = sorted(glob.glob(os.path.join(args.test_path, '*.[jp][pn]g')))
    bs = args.batch_size or (16 if device.type == 'cuda' else 1)
    for s in range(0, len(paths), bs):
        chunk = paths[s:s + bs]
        for q in chunk:
            print(os.path.basename(q))
        x = faces_to_tensor([imread_bgr(q) for q in chunk], device)        # img2tensor(img / 255.) + normalize(0.5, 0.5)
        with torch.no_grad():
            out = tensor_to_faces(vqgan(x)[0])                             # tensor2img(output, min_max=[-1, 1])
        for q, img in zip(chunk, out):
            imwrite(img, os.path.splitext(os.path.join(save_root, os.path.basename(q)))[0] + '.png')
    print(f'\nAll results are saved in {save_root}')
    return 0


if __name__ == '__main__':
    sys.exit(main())
