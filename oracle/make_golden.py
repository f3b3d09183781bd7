"""Generate tests/golden/*.npz from the REFERENCE's own arch files (build container only).

TEST INFRASTRUCTURE ONLY.  Run:  python -m oracle.make_golden
Needs /root/reference (absent on the GPU box -- the outputs are committed).

What it pins
  * seed-0 random-init weights (the reference ships no checkpoint):
    per-key sha256 + the SURVEY 8(c) known-answer values;
  * CodeFormer.forward of the reference on the seeded 512x512 input
    (restoration config, w in {0, 0.5, 1.0}, adain=True) and the inpainting config
    (codebook 512, 3 fuse levels, w=1, adain=False);
  * that oracle/codeformer_oracle.py reproduces those outputs (max |diff| recorded
    in tests/golden/oracle_vs_reference.json and asserted by the CPU tests).
"""
import hashlib
import json
import os
import sys
import time

import numpy as np
import torch

from oracle import codeformer_oracle as O
from oracle import ref_loader
from oracle.synth import seeded_input, seeded_randn, synth_state_dict

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = os.path.join(os.path.dirname(HERE), 'tests', 'golden')


def sd_digest(sd):
    return {k: hashlib.sha256(v.detach().cpu().contiguous().numpy().tobytes()).hexdigest()[:16]
            for k, v in sd.items()}


def top2_gap(logits):
    t = torch.topk(logits, 2, dim=-1).values
    return (t[..., 0] - t[..., 1])


def main():
    os.makedirs(GOLD, exist_ok=True)
    torch.set_num_threads(os.cpu_count())
    reg, vq, cf, _ = ref_loader.load_reference()
    report = {'torch': torch.__version__, 'threads': torch.get_num_threads()}

    # ---------------- restoration config ----------------
    torch.manual_seed(0)
    net = reg.ARCH_REGISTRY.get('CodeFormer')(dim_embd=512, codebook_size=1024, n_head=8, n_layers=9,
                                              connect_list=['32', '64', '128', '256']).eval()
    sd = {k: v.detach() for k, v in net.state_dict().items()}
    abs_sum = float(sum(v.double().abs().sum() for v in sd.values()))
    report['sd_abs_sum'] = abs_sum
    assert abs(abs_sum - 1218901.639128) < 1e-3, abs_sum
    x = seeded_input(1)
    assert abs(float(x[0, 0, 0, 0]) - (-0.94204152)) < 1e-6

    gold = {}
    for w in (0.5, 0.0, 1.0):
        t0 = time.time()
        with torch.no_grad():
            out, logits, lq = net(x, w=w, adain=True)
        dt = time.time() - t0
        idx = torch.topk(torch.softmax(logits, dim=2), 1, dim=2)[1].view(1, -1)
        o_out, o_logits, o_lq, o_idx = O.codeformer_forward(x, sd, w=w, adain_flag=True, return_idx=True)
        tag = f'w{w}'
        report[tag] = {
            'ref_seconds': dt,
            'out_mean': float(out.mean()), 'out_absmean': float(out.abs().mean()),
            'oracle_vs_ref_out': float((o_out - out).abs().max()),
            'oracle_vs_ref_logits': float((o_logits - logits).abs().max()),
            'oracle_vs_ref_lq': float((o_lq - lq).abs().max()),
            'oracle_idx_equal': bool(torch.equal(o_idx, idx)),
            'idx_sha': hashlib.sha256(idx.numpy().astype('<i8').tobytes()).hexdigest()[:16],
            'argmax_equals_topk_softmax': bool(torch.equal(logits.argmax(-1), idx)),
            'min_gap': float(top2_gap(logits).min()),
        }
        print(tag, report[tag], flush=True)
        if w == 0.5:
            gold.update(out=out.numpy(), logits=logits.numpy(), lq_feat=lq.numpy(), idx=idx.numpy(),
                        gap=top2_gap(logits).numpy())
        else:
            gold[f'out_{tag}_sub'] = out[:, :, ::4, ::4].contiguous().numpy()
    assert report['w0.5']['idx_sha'] == '7d2fd85619ab8528', report['w0.5']['idx_sha']
    np.savez_compressed(os.path.join(GOLD, 'restoration_seed0_face0.npz'), **gold)
    with open(os.path.join(GOLD, 'state_dict_seed0_digest.json'), 'w') as f:
        json.dump(sd_digest(sd), f, indent=0, sort_keys=True)

    # second face (batch index 1 of the seeded batch-16 input of config 2) -- indices + subsampled out
    xb = seeded_input(16)
    with torch.no_grad():
        out1, logits1, lq1 = net(xb[1:2], w=0.5, adain=True)
    np.savez_compressed(os.path.join(GOLD, 'restoration_seed0_b16_face1.npz'),
                        out_sub=out1[:, :, ::4, ::4].contiguous().numpy(), logits=logits1.numpy(),
                        idx=logits1.argmax(-1).numpy(), gap=top2_gap(logits1).numpy(),
                        lq_feat=lq1.numpy())

    # ---------------- per-block fixtures: reference modules with synthetic weights -------------
    # (weights are regenerated from {name: shape} + seed by oracle/synth.py, only I/O is stored)
    mods = {
        'res': vq.ResBlock(64, 128), 'attn': vq.AttnBlock(512),
        'tl': cf.TransformerSALayer(embed_dim=512, nhead=8, dim_mlp=1024, dropout=0.0),
        'fuse': cf.Fuse_sft_block(128, 128), 'down': vq.Downsample(64), 'up': vq.Upsample(64),
    }
    shapes = {p + '.' + k: list(v.shape) for p, m in mods.items() for k, v in m.state_dict().items()}
    bsd = synth_state_dict(shapes, 7)
    for p, m in mods.items():
        m.load_state_dict({k[len(p) + 1:]: v for k, v in bsd.items() if k.startswith(p + '.')})
        m.eval()
    xr = seeded_randn((1, 64, 32, 32), 71)
    xa = seeded_randn((1, 512, 16, 16), 72)
    xt = seeded_randn((256, 2, 512), 73)
    pos = seeded_randn((256, 1, 512), 74).repeat(1, 2, 1)
    xe, xd = seeded_randn((1, 128, 32, 32), 75), seeded_randn((1, 128, 32, 32), 76)
    xs = seeded_randn((1, 64, 32, 32), 77)
    with torch.no_grad():
        blk_out = {
            'res': mods['res'](xr), 'attn': mods['attn'](xa), 'tl': mods['tl'](xt, query_pos=pos),
            'fuse': mods['fuse'](xe, xd, 0.7), 'down': mods['down'](xs), 'up': mods['up'](xs),
            'adain': cf.adaptive_instance_normalization(xe, xd),
        }
    chk = {
        'res': O.res_block(xr, bsd, 'res'), 'attn': O.attn_block(xa, bsd, 'attn'),
        'tl': O.transformer_layer(xt, pos, bsd, 'tl', 8), 'fuse': O.fuse_sft(xe, xd, 0.7, bsd, 'fuse'),
        'down': O.downsample(xs, bsd, 'down'), 'up': O.upsample(xs, bsd, 'up'), 'adain': O.adain(xe, xd),
    }
    report['blocks'] = {k: float((chk[k] - blk_out[k]).abs().max()) for k in chk}
    print('blocks', report['blocks'], flush=True)
    np.savez_compressed(os.path.join(GOLD, 'blocks_seed7.npz'), **{'out_' + k: v.numpy() for k, v in blk_out.items()})
    with open(os.path.join(GOLD, 'blocks_seed7_shapes.json'), 'w') as f:
        json.dump(shapes, f, sort_keys=True)

    # VectorQuantizer.forward (next-row f2)
    torch.manual_seed(11)
    q = vq.VectorQuantizer(1024, 256, 0.25).eval()
    z = torch.randn(2, 256, 16, 16) * 1e-3
    with torch.no_grad():
        zq, _, st = q(z)
    ozq, oidx, _ = O.vq_nearest(z, q.embedding.weight.detach())
    report['vq'] = {'idx_equal': bool(torch.equal(oidx, st['min_encoding_indices'].view(-1))),
                    'zq_diff': float((ozq - zq).abs().max())}
    np.savez_compressed(os.path.join(GOLD, 'vq_seed11.npz'), z=z.numpy(), codebook=q.embedding.weight.detach().numpy(),
                        idx=st['min_encoding_indices'].view(-1).numpy(), zq=zq.numpy())

    # ---------------- inpainting config (inference_inpainting.py:45-46,73) ----------------
    torch.manual_seed(0)
    net_i = reg.ARCH_REGISTRY.get('CodeFormer')(dim_embd=512, codebook_size=512, n_head=8, n_layers=9,
                                                connect_list=['32', '64', '128']).eval()
    sd_i = {k: v.detach() for k, v in net_i.state_dict().items()}
    with torch.no_grad():
        out_i, logits_i, lq_i = net_i(x, w=1, adain=False)
    oi = O.codeformer_forward(x, sd_i, w=1, adain_flag=False, connect_list=('32', '64', '128'), return_idx=True)
    report['inpaint'] = {'oracle_vs_ref_out': float((oi[0] - out_i).abs().max()),
                         'oracle_vs_ref_logits': float((oi[1] - logits_i).abs().max()),
                         'idx_equal': bool(torch.equal(oi[3], logits_i.argmax(-1))),
                         'min_gap': float(top2_gap(logits_i).min())}
    print('inpaint', report['inpaint'], flush=True)
    np.savez_compressed(os.path.join(GOLD, 'inpaint_seed0_face0.npz'),
                        out_sub=out_i[:, :, ::4, ::4].contiguous().numpy(), logits=logits_i.numpy(),
                        idx=logits_i.argmax(-1).numpy(), gap=top2_gap(logits_i).numpy())
    with open(os.path.join(GOLD, 'state_dict_inpaint_seed0_digest.json'), 'w') as f:
        json.dump(sd_digest(sd_i), f, indent=0, sort_keys=True)

    with open(os.path.join(GOLD, 'oracle_vs_reference.json'), 'w') as f:
        json.dump(report, f, indent=1, sort_keys=True)
    print('done')


if __name__ == '__main__':
    sys.exit(main())
