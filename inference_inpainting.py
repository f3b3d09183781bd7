"""Face inpainting on aligned 512x512 masked faces -- MI355X-native drop-in for the reference's inference_inpainting.py
(same flags and result tree; reference lines 14-91).  Network: CodeFormer(codebook_size=512, 3 fuse levels), w=1,
adain=False; the white-brush mask composite runs on the device.  Faces are processed in batches."""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from basicsr.utils import imwrite  # noqa: E402
from basicsr.utils.misc import get_device  # noqa: E402
from codeformer_amd import cli  # noqa: E402

pretrain_model_url = 'https://github.com/sczhou/CodeFormer/releases/download/v0.1.0/codeformer_inpainting.pth'


def main(argv=None):
    p = argparse.ArgumentParser()
    p.add_argument('-i', '--input_path', type=str, default='./inputs/masked_faces', help='Input image or folder.')
    p.add_argument('-o', '--output_path', type=str, default=None, help='Output folder. Default: results/<input_name>')
    p.add_argument('--suffix', type=str, default=None, help='Suffix of the restored faces. Default: None')
    p.add_argument('--batch_size', type=int, default=None)
    p.add_argument('--device', type=str, default=None)
    p.add_argument('--random_init_seed', type=int, default=None)
    args = p.parse_args(argv)
    device = torch.device(args.device) if args.device else get_device()
    print('[NOTE] The input face images should be aligned and cropped to a resolution of 512x512.')
    paths, folder = cli.list_images(args.input_path)
    result_root = args.output_path or ('results/test_inpainting_img' if folder is None else f'results/{folder}')
    net = cli.build_codeformer(device, 'codeformer_inpainting.pth', pretrain_model_url, 512, ['32', '64', '128'],
                               args.random_init_seed)
    bs = args.batch_size or (16 if device.type == 'cuda' else 1)
    for s in range(0, len(paths), bs):
        chunk = paths[s:s + bs]
        for j, q in enumerate(chunk):
            print(f'[{s + j + 1}/{len(paths)}] Processing: {os.path.basename(q)}')
        x = cli.faces_to_tensor(cli.load_faces(chunk, require=(512, 512)), device)
        try:
            with torch.no_grad():
                faces = cli.tensor_to_faces(cli.inpaint_composite(x, net(x, w=1, adain=False)[0]))
        except Exception as error:  # reference behaviour: report and return the input face
            print(f'\tFailed inference for CodeFormer: {error}')
            faces = cli.tensor_to_faces(x)
        for q, face in zip(chunk, faces):
            base = os.path.splitext(os.path.basename(q))[0]
            if args.suffix is not None:
                base = f'{base}_{args.suffix}'
            imwrite(face.astype('uint8'), os.path.join(result_root, f'{base}.png'))
    print(f'\nAll results are saved in {result_root}')


if __name__ == '__main__':
    main()
