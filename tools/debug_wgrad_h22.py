"""Where the flat-pipeline half weight gradient goes wrong on 2 x 2 blocks (B2M_WGRAD_H_PIPE_SMALL=1 forces it): structured operands.
x = 1, dy = 1: dW[k][ci][co] = the pair count of offset k for every (ci, co) -- which elements differ, by how many pairs."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import torch
from box2mask_amd import functional as F_, half_train as HT, _lib
from box2mask_amd.sparse import CoordinateManager
from test_gpu_ops import _scene
b = _scene()
m = CoordinateManager(b['vox_coords']); m.ensure_level(2)
HT.loss_scale[0] = 1.0
torch.set_printoptions(linewidth=250, sci_mode=False)
for (lvl, cin, cout) in ((0, 32, 32), (0, 64, 64)):
    rb = m.rulebook_same(lvl, 3); n = m.n(lvl)
    for name, x, dy in (('ones', torch.ones(n, cin), torch.ones(n, cout)),
                        ('x=1, dy=col', torch.ones(n, cin), torch.arange(cout).float()[None].repeat(n, 1)),
                        ('x=col, dy=1', torch.arange(cin).float()[None].repeat(n, 1), torch.ones(n, cout)),
                        ('x=row%7, dy=1', (torch.arange(n) % 7).float()[:, None].repeat(1, cin), torch.ones(n, cout)),
                        ('x=1, dy=row%7', torch.ones(n, cin), (torch.arange(n) % 7).float()[:, None].repeat(1, cout))):
        x = x.cuda().half(); dy = dy.cuda().half()
        out = {}
        for small in ('0', '1'):
            os.environ['B2M_WGRAD_H_PIPE_SMALL'] = small; _lib.reload_env()
            dw = torch.zeros(27, cin, cout, device='cuda')
            HT._wgrad_h(x, dy, rb, 27, dw, 0, 1.0)
            torch.cuda.synchronize()
            out[small] = dw.cpu()
        d = out['1'] - out['0']
        bad = (d != 0)
        print('L%d %d->%d %-14s wrong elements %d of %d; offsets %s' % (lvl, cin, cout, name, int(bad.sum()), bad.numel(),
              [int(k) for k in bad.any(2).any(1).nonzero().flatten()][:27]))
        if bad.any():
            k = int(bad.any(2).any(1).nonzero().flatten()[0])
            print('   offset %d: rows (ci) wrong %s' % (k, [int(v) for v in bad[k].any(1).nonzero().flatten()]))
            print('   offset %d: cols (co) wrong %s' % (k, [int(v) for v in bad[k].any(0).nonzero().flatten()]))
            print('   plain kernel [k, :4, :8]\n', out['0'][k, :4, :8]); print('   flow kernel [k, :4, :8]\n', out['1'][k, :4, :8])
            print('   ratio flow / plain over the offsets (element [0, 0]):', (out['1'][:, 0, 0] / out['0'][:, 0, 0]).numpy().round(3))
            print('   ratio flow / plain over the offsets (element [16, 16]):', (out['1'][:, 16, 16] / out['0'][:, 16, 16]).numpy().round(3))
