"""Weight-file resolution (API of the reference's basicsr/utils/download_util.py:69-94).

There is no network on the target boxes: an already-present file is returned as-is (the reference's behaviour
for cached files), anything else raises instead of attempting a download.
"""
import os
from urllib.parse import urlparse


def load_file_from_url(url, model_dir=None, progress=True, file_name=None):
    if model_dir is None:
        model_dir = os.path.join(os.path.expanduser('~'), '.cache', 'codeformer_amd')
    filename = file_name or os.path.basename(urlparse(url).path)
    cached = os.path.abspath(os.path.join(model_dir, filename))
    if os.path.exists(cached):
        return cached
    raise FileNotFoundError(f'{cached} is not present and downloads are disabled (source: {url})')
