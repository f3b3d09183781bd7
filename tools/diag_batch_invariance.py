import sys, os, torch
sys.path.insert(0, '/root/repo')
from codeformer_amd import ops
torch.manual_seed(0)
def chk(name, a, b):
    print(name, 'bitwise equal' if torch.equal(a, b) else f'DIFF max {float((a-b).abs().max()):.3e}', flush=True)
for (cin, cout, H) in ((64, 64, 512), (128, 128, 256), (128, 128, 128), (256, 256, 64), (256, 256, 32)):
    x = torch.randn(4, H, H, cin, device='cuda')
    w = torch.randn(cout, cin, 3, 3, device='cuda') * 0.05
    b = torch.randn(cout, device='cuda')
    pw = ops.pack_weight(w, b)
    xp = torch.nn.functional.pad(x, (0, 0, 0, 1, 0, 1))
    y4 = ops.conv2d(x, pw, stride=2, emit_stats=True)
    y1 = ops.conv2d(x[:1].contiguous(), pw, stride=2, emit_stats=True)
    chk(f's2 {cin}->{cout}@{H} out', y4[:1], y1)
    s4, s1 = y4._cf_stats, y1._cf_stats
    chk(f's2 {cin}->{cout}@{H} stats', s4.part.view(4, -1)[:1], s1.part.view(1, -1))
    g = torch.ones(cout, device='cuda'); bb = torch.zeros(cout, device='cuda')
    t4 = ops.groupnorm_tables([y4], g, bb); t1 = ops.groupnorm_tables([y1], g, bb)
    chk('   gn tables', t4[0][:1], t1[0])
    for code, nm in ((ops.SPLIT, 'split'), (ops.WSPLIT, 'wsplit'), (ops.WINOGRAD, 'wino')):
        pw2 = ops.pack_weight(w, b, bf16=code)
        z4 = ops.conv2d(x, pw2, emit_stats=True, prologue=ops.PRO_AFFINE_SWISH, scale=torch.ones(4, cin, device='cuda'), shift=torch.zeros(4, cin, device='cuda'))
        z1 = ops.conv2d(x[:1].contiguous(), pw2, emit_stats=True, prologue=ops.PRO_AFFINE_SWISH, scale=torch.ones(1, cin, device='cuda'), shift=torch.zeros(1, cin, device='cuda'))
        chk(f'   {nm} out', z4[:1], z1)
        chk(f'   {nm} stats', z4._cf_stats.part.view(4, -1)[:1], z1._cf_stats.part.view(1, -1))
x = torch.randn(4, 16, 16, 512, device='cuda'); w = torch.randn(512, 512, 3, 3, device='cuda') * 0.02; b = torch.randn(512, device='cuda')
for code, nm in ((ops.WSPLIT, 'wsplit16'), (ops.WINOGRAD, 'wino16')):
    pw2 = ops.pack_weight(w, b, bf16=code)
    chk(nm, ops.conv2d(x, pw2)[:1], ops.conv2d(x[:1].contiguous(), pw2))
q = torch.randn(4 * 256, 1536, device='cuda')
o4 = ops.attention(q[:, :512], q[:, 512:1024], q[:, 1024:], 4, 1, 512, 512 ** -0.5) if hasattr(ops, 'attention') else None
if o4 is not None:
    o1 = ops.attention(q[:256, :512], q[:256, 512:1024], q[:256, 1024:], 1, 1, 512, 512 ** -0.5)
    chk('attn512', o4[:256], o1)
x = torch.randn(4, 3, 512, 512, device='cuda'); w = torch.randn(64, 3, 3, 3, device='cuda'); b = torch.randn(64, device='cuda')
pw = ops.pack_weight(w, b)
f4 = ops.conv2d(x, pw, in_nchw=True, emit_stats=True); f1 = ops.conv2d(x[:1].contiguous(), pw, in_nchw=True, emit_stats=True)
chk('first conv out', f4[:1], f1); chk('first conv stats', f4._cf_stats.part.view(4, -1)[:1], f1._cf_stats.part.view(1, -1))
