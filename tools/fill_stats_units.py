import sys, numpy as np, runpy, os
sys.path.insert(0,'/root/repo')
src = open('/root/repo/tools/fill_stats.py').read()
head = src.split("for lvl in range(4):")[0]
exec(head)
for lvl in range(4):
    ts = 1<<lvl
    nbr = np.asarray(S.kernel_map_same(c, 3, ts))
    K, N = nbr.shape
    for T in (64, 128, 256, 512):
        nt = (N + T - 1)//T
        pad = nt*T - N
        v = np.concatenate([nbr >= 0, np.zeros((K,pad),bool)],1).reshape(K, nt, T).sum(2)
        act = v > 0
        g = (v + 15)//16
        units = (v + 63)//64     # visits of <= 4 groups
        # group count of each unit: full units have 4, last has ceil((v%64)/16) or 4
        lastg = np.where(v % 64 == 0, 4, ((v % 64) + 15)//16)
        lastg = np.where(v == 0, 0, lastg)
        nfull = np.where(v>0, units - 1, 0)
        hist = np.zeros(5)
        hist[4] += nfull.sum()
        for G in range(1,5): hist[G] += (lastg[act] == G).sum()
        full64 = (v == T).sum() if T == 64 else 0
        print('level', lvl, 'N', N, 'T', T, 'useful %.3f'%(v.sum()/(16*g.sum())), 'units/row %.3f'%(units.sum()/N), 'mfma groups/unit %.2f'%(g.sum()/units.sum()),
              'unit G hist', np.round(hist[1:]/hist.sum(),3), 'mfma share in G=1 units %.3f'%(hist[1]/g.sum()), 'full64 share of visits %.3f'%(full64/act.sum()), 'units per T-tile %.1f'%(units.sum()/nt))
    c = S.stride_coords(c, ts)[0].astype(np.int64)
    c = c[hilbert(c) if ORDER == 'hilbert' else morton(c)]
