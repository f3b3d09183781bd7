"""BASELINE config 1: the --has_aligned entrypoint on CPU with random-init weights (plumbing, no GPU)."""
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _faces(d, n):
    from PIL import Image
    rng = np.random.default_rng(0)
    os.makedirs(d, exist_ok=True)
    for i in range(n):
        Image.fromarray(rng.integers(0, 256, (512, 512, 3), dtype=np.uint8)).save(os.path.join(d, f'f{i}.png'))


def test_has_aligned_cpu_plumbing(tmp_path):
    src, dst = tmp_path / 'cropped_faces', tmp_path / 'out'
    _faces(str(src), 2)
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'inference_codeformer.py'), '--has_aligned', '-i', str(src), '-o',
                        str(dst), '-w', '0.5', '--device', 'cpu', '--random_init_seed', '0', '--batch_size', '2',
                        '--suffix', 'r'], capture_output=True, text=True, timeout=600, cwd=str(tmp_path))
    assert r.returncode == 0, r.stdout + r.stderr
    assert 'Failed inference' not in r.stdout                       # the reference's swallowed-exception marker (F7)
    assert sorted(os.listdir(dst / 'restored_faces')) == ['f0_r.png', 'f1_r.png']
    from PIL import Image
    assert Image.open(dst / 'restored_faces' / 'f0_r.png').size == (512, 512)
    assert 'All results are saved in' in r.stdout


def test_missing_checkpoint_is_an_error_without_opt_in(tmp_path):
    src = tmp_path / 'faces'
    _faces(str(src), 1)
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'inference_codeformer.py'), '--has_aligned', '-i', str(src),
                        '--device', 'cpu'], capture_output=True, text=True, timeout=300, cwd=str(tmp_path))
    assert r.returncode != 0 and 'codeformer.pth' in (r.stdout + r.stderr)


def test_whole_image_path_needs_the_device(tmp_path):
    """Whole images: detection runs on the host, but crop warp and paste-back are HIP kernels -- on a CPU-only box the entrypoint says
    so instead of silently doing something else; .mp4 input says it needs ffmpeg."""
    src = tmp_path / 'faces'
    _faces(str(src), 1)
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'inference_codeformer.py'), '-i', str(src), '--device', 'cpu',
                        '--random_init_seed', '0'], capture_output=True, text=True, timeout=300, cwd=str(tmp_path))
    assert r.returncode != 0 and 'ROCm device' in (r.stdout + r.stderr)
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'inference_codeformer.py'), '-i', str(tmp_path / 'clip.mp4'), '--device', 'cpu'],
                       capture_output=True, text=True, timeout=300, cwd=str(tmp_path))
    assert r.returncode != 0 and 'ffmpeg' in (r.stdout + r.stderr)


def test_inpainting_entrypoint_cpu(tmp_path):
    """BASELINE config 5 plumbing: only pure-white pixels are replaced (inference_inpainting.py:68-74)."""
    from PIL import Image
    rng = np.random.default_rng(1)
    src = tmp_path / 'masked_faces'
    os.makedirs(src)
    a = rng.integers(0, 250, (512, 512, 3), dtype=np.uint8)
    a[100:200, 150:300] = 255
    Image.fromarray(a).save(src / 'm0.png')
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'inference_inpainting.py'), '-i', str(src), '-o', str(tmp_path / 'o'),
                        '--device', 'cpu', '--random_init_seed', '0'], capture_output=True, text=True, timeout=600, cwd=str(tmp_path))
    assert r.returncode == 0 and 'Failed inference' not in r.stdout, r.stdout + r.stderr
    b = np.asarray(Image.open(tmp_path / 'o' / 'm0.png'))
    m = (a == 255).all(-1)
    assert np.array_equal(a[~m], b[~m]) and (a[m] != b[m]).any()


def test_colorization_entrypoint_cpu(tmp_path):
    src = tmp_path / 'gray_faces'
    _faces(str(src), 1)
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'inference_colorization.py'), '-i', str(src), '-o', str(tmp_path / 'o'),
                        '--device', 'cpu', '--random_init_seed', '0', '--suffix', 'c'], capture_output=True, text=True, timeout=600,
                       cwd=str(tmp_path))
    assert r.returncode == 0 and 'Failed inference' not in r.stdout, r.stdout + r.stderr
    assert os.listdir(tmp_path / 'o') == ['f0_c.png']


def test_vqgan_reconstruction_script_cpu(tmp_path):
    """scripts/inference_vqgan.py (the caller of VectorQuantizer.forward, reference scripts/inference_vqgan.py:12-59): CPU plumbing with
    seeded weights; a missing checkpoint without the opt-in is an error."""
    src = tmp_path / 'ffhq_512'
    _faces(str(src), 2)
    script = os.path.join(ROOT, 'scripts', 'inference_vqgan.py')
    r = subprocess.run([sys.executable, script, '-i', str(src), '-o', str(tmp_path / 'rec') + '/', '--device', 'cpu', '--random_init_seed', '0'],
                       capture_output=True, text=True, timeout=600, cwd=str(tmp_path))
    assert r.returncode == 0, r.stdout + r.stderr
    assert sorted(os.listdir(tmp_path / 'rec')) == ['f0.png', 'f1.png'] and 'All results are saved in' in r.stdout
    from PIL import Image
    assert Image.open(tmp_path / 'rec' / 'f0.png').size == (512, 512)
    r = subprocess.run([sys.executable, script, '-i', str(src), '-o', str(tmp_path / 'rec2'), '--device', 'cpu'], capture_output=True, text=True,
                       timeout=300, cwd=str(tmp_path))
    assert r.returncode != 0 and 'net_g.pth' in (r.stdout + r.stderr)
