"""ORACLE (test infrastructure; never imported by the product).

CPU restatement of SelectionNet.forward (/root/reference/models/detection_net.py:234-364) and of
BasicBlock.forward (/root/reference/models/resnet.py:70-83) on top of oracle/sparse_ref.py, written
as a pure function of a state dict with the reference's parameter names, so it can be
run against the product's weights.  torch CPU autograd provides the backward pass.

PARITY UNPINNED at the sparse-engine level (see the header of oracle/sparse_ref.py): the layer
schedule, channel plan and head structure follow the reference source line by line, but the numeric
behaviour of each sparse operator is pinned only by the independent checks in
tests/test_oracle_sparse.py, not by MinkowskiEngine itself.
"""
from __future__ import annotations

import torch

from . import sparse_ref as S


def _bn(p, name, x, training, stats_out=None):
    return S.batch_norm(x, p[name + '.bn.weight'], p[name + '.bn.bias'],
                        p[name + '.bn.running_mean'].clone(), p[name + '.bn.running_var'].clone(), training)


def _block(p, name, x, nbr, training):
    """BasicBlock (resnet.py:70-83): conv3-BN-ReLU-conv3-BN (+1x1conv-BN residual) add ReLU."""
    out = S.conv_nbr(x, p[name + '.conv1.kernel'], nbr)
    out = torch.relu(_bn(p, name + '.norm1', out, training))
    out = S.conv_nbr(out, p[name + '.conv2.kernel'], nbr)
    out = _bn(p, name + '.norm2', out, training)
    if (name + '.downsample.0.kernel') in p:
        res = S.conv_nbr(x, p[name + '.downsample.0.kernel'], None)
        res = _bn(p, name + '.downsample.1', res, training)
    else:
        res = x
    return torch.relu(out + res)


def _layer(p, name, x, nbr, training, n_blocks):
    for b in range(n_blocks):
        x = _block(p, '%s.%d' % (name, b), x, nbr, training)
    return x


def _head(p, name, x, training):
    """mlp_head (detection_net.py:170-194): conv1x1(+bias)-ReLU-BN, conv1x1-ReLU-BN, conv1x1."""
    x = torch.relu(S.conv_nbr(x, p[name + '.0.kernel'], None, p[name + '.0.bias']))
    x = _bn(p, name + '.2', x, training)
    x = torch.relu(S.conv_nbr(x, p[name + '.3.kernel'], None, p[name + '.3.bias']))
    x = _bn(p, name + '.5', x, training)
    return S.conv_nbr(x, p[name + '.6.kernel'], None, p[name + '.6.bias'])


HEAD_ATTR = {'mlp_offsets': 'mlp_offsets', 'mlp_bounds': 'mlp_bounds', 'mlp_bb_scores': 'mlp_score',
             'mlp_center_scores': 'mlp_center_score', 'mlp_semantics': 'mlp_semantics',
             'mlp_per_vox_semantics': 'mlp_per_vox_semantics'}


def forward(p, coords, feats, pooling_ids, cfg, training=True, hier: S.Hierarchy | None = None, n_segments=None,
            return_trunk=False, trace=None):
    """p: dict name -> CPU tensor (the product's state_dict moved to the CPU).  Returns head -> tensor."""
    h = hier if hier is not None else S.Hierarchy(coords)
    L = cfg.layers
    x = feats

    def cbr(conv, bn, x, nbr):
        return torch.relu(_bn(p, bn, S.conv_nbr(x, p[conv + '.kernel'], nbr), training))

    def T(name, t):
        if trace is not None:
            trace[name] = t
        return t

    out_p1 = T('out_p1', cbr('conv0p1s1', 'bn0', x, h.k_first()))
    enc = [out_p1]
    names = [('conv1p1s2', 'bn1', 'block1'), ('conv2p2s2', 'bn2', 'block2'), ('conv3p4s2', 'bn3', 'block3'),
             ('conv4p8s2', 'bn4', 'block4'), ('added_conv1p16s2', 'added_bn1', 'added_block1'),
             ('added_conv2p32s2', 'added_bn2', 'added_block2'), ('added_conv3p64s2', 'added_bn3', 'added_block3')]
    out = out_p1
    for l, (c, b, blk) in enumerate(names):          # level l -> l+1
        out = T('down%d' % (l + 1), cbr(c, b, out, h.down(l)))
        out = T(blk, _layer(p, blk, out, h.k3(l + 1), training, L))
        enc.append(out)
    ups = [('added_convtr4p128s2', 'added_bntr4', 'added_block4'), ('added_convtr5p64s2', 'added_bntr5', 'added_block5'),
           ('added_convtr6p32s2', 'added_bntr6', 'added_block6'), ('convtr4p16s2', 'bntr4', 'block5'),
           ('convtr5p8s2', 'bntr5', 'block6'), ('convtr6p4s2', 'bntr6', 'block7'), ('convtr7p2s2', 'bntr7', 'block8')]
    for j, (c, b, blk) in enumerate(ups):            # level 7-j -> 6-j
        l = 6 - j
        out = T('up%d' % l, cbr(c, b, out, h.up(l)))
        out = torch.cat([out, enc[l]], 1)            # ME.cat(upsampled, skip): detection_net.py:286-336
        out = T(blk, _layer(p, blk, out, h.k3(l), training, L))
    trunk = out
    outputs = {}
    per_vox = any('per_vox' in hd for hd in cfg.network_heads)
    if per_vox:
        outputs['vox_feats'] = trunk
    if cfg.do_segment_pooling:
        ids = torch.as_tensor(pooling_ids).long()
        n_seg = int(ids.max()) + 1 if n_segments is None else n_segments
        out = S.segment_pool(out, ids, n_seg, 'max' if cfg.max_pool_segments_detection_net else 'avg')
    for hd in cfg.network_heads:
        src = trunk if (per_vox and 'per_vox' in hd) else out
        y = _head(p, HEAD_ATTR[hd], src, training)
        if cfg.mlp_bounds_relu and hd == cfg.mlp_bounds:
            y = torch.relu(y)
        outputs[hd] = y
    if return_trunk:
        outputs['_trunk'] = trunk
    return outputs
