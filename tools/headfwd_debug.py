import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "opensearch-sparse-model-tuning-sample_amd")]
from sparse_hip import ops
H, V = 128, 300
for (B, S) in [(1, 32), (2, 32), (3, 32), (4, 32), (5, 32), (2, 64), (3, 64), (5, 64), (4, 128)]:
    g = torch.Generator().manual_seed(1)
    t = torch.randn(B * S, H, generator=g).to(torch.bfloat16).cuda()
    E = (torch.randn(384, H, generator=g) * 0.3).to(torch.bfloat16).cuda()
    bias = (torch.randn(V, generator=g) * 0.5).cuda()
    mask = torch.ones(B, S, dtype=torch.uint8)
    for b in range(1, B):
        mask[b, S - 3 * b - 2:] = 0
    rep, am = ops.sparse_head_fwd(t, E, bias, mask.cuda().reshape(-1), B, S, V, False)
    torch.cuda.synchronize()
    lg = (t.float() @ E[:V].float().t()).view(B, S, V)
    lg = lg.masked_fill(mask.cuda()[:, :, None] == 0, -1e30)
    want = torch.log1p(torch.relu(lg.max(1).values + bias))
    errs = (rep - want).abs().amax(1).cpu().tolist()
    print(B, S, "nsteps", B * S // 32, ["%.1e" % e for e in errs])
