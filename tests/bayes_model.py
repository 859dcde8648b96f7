"""Test harness for the Bayes filter: the part of Memory the filter talks to (a pose graph and Memory::getNeighborsId), in Python.

graph: signatures 1..n in odometry order (neighbour links i <-> i+1) plus loop-closure links.  neighbors() is the traversal of
Memory::getNeighborsId(id, maxGraphDepth, 0, incrementMarginOnLoop=false, ignoreLoopIds=false, ...) (reference Memory.cpp:1703-1890):
breadth first, a loop-closure link does not increase the margin, margins 0 .. maxGraphDepth - 1, the signature itself at margin 0.
"""
import collections

import numpy as np

import oracle as O


def prediction_lc_as_parsed(values):
    """BayesFilter::setPredictionLC parses the string with uStr2Float (float) and stores doubles (BayesFilter.cpp:88-92)."""
    return np.asarray(values, dtype=np.float32).astype(np.float64)


DEFAULT_LC = prediction_lc_as_parsed(O.DEFAULT_PREDICTION_LC)


class Graph:
    def __init__(self, n, loops=()):
        self.n = n
        self.odom = collections.defaultdict(set)
        self.loop = collections.defaultdict(set)
        for i in range(1, n):
            self.odom[i].add(i + 1)
            self.odom[i + 1].add(i)
        for a, b in loops:
            if a != b:
                self.loop[a].add(b)
                self.loop[b].add(a)

    def neighbors(self, sid, max_depth):
        """{id: margin}: 0-1 breadth-first search (loop links cost 0, odometry links 1), margins < max_depth."""
        dist = {sid: 0}
        dq = collections.deque([sid])
        while dq:
            u = dq.popleft()
            d = dist[u]
            for v in self.loop[u]:
                if v not in dist or dist[v] > d:
                    dist[v] = d
                    dq.appendleft(v)
            if d + 1 < max_depth:
                for v in self.odom[u]:
                    if v not in dist or dist[v] > d + 1:
                        dist[v] = d + 1
                        dq.append(v)
        return dist


def random_graph(n, n_loops, rng):
    loops = []
    for _ in range(n_loops):
        a = int(rng.integers(20, n + 1)) if n > 20 else int(rng.integers(1, n + 1))
        b = int(rng.integers(1, max(a - 10, 2)))
        loops.append((a, b))
    return Graph(n, loops)


def csr_lists(graph, ids, max_depth, keep=None):
    """(offsets, nbr ids, margins) of the neighbour lists of `ids`, ascending neighbour id (std::map order); keep: filter on ids."""
    off = [0]
    nbr, mg = [], []
    for s in ids:
        d = graph.neighbors(int(s), max_depth)
        for k in sorted(d):
            if keep is None or keep(k):
                nbr.append(k)
                mg.append(d[k])
        off.append(len(nbr))
    return np.asarray(off, np.int64), np.asarray(nbr, np.int32), np.asarray(mg, np.int32)


def random_adjusted(m, rng):
    """An adjusted likelihood as Rtabmap::adjustLikelihood leaves it: 1 for most signatures, a few above, the virtual place first."""
    like = np.ones(m, np.float32)
    k = max(1, m // 50)
    hot = rng.choice(np.arange(1, m), size=min(k, m - 1), replace=False) if m > 1 else np.zeros(0, np.int64)
    like[hot] = (1.0 + rng.gamma(2.0, 2.0, size=hot.shape[0])).astype(np.float32)
    like[0] = np.float32(1.0 + rng.random() * 2.0)
    return like


# ---------------------------------------------------------------------------------------------------------------------------------
# The device algorithm (rtabmap_amd/csrc/bayes.hip) step by step in numpy: per-column scalars from the neighbour lists, rows gathered
# from the (symmetric) lists with the "all other places" fill carried as one sum, double accumulation, one rounding to float.  Runs
# without a GPU, so the CPU suite checks the ALGORITHM against the oracle; the GPU suite checks the kernels.
f32=np.float32
def params(lc, vp):
    lc=np.asarray(lc,np.float64); total=f32(0)
    eps=f32(0)
    for j,v in enumerate(lc):
        total=f32(np.float64(total)+v)
        if j==0 or v<np.float64(eps): eps=f32(v)
    return dict(lc=lc.astype(f32), lc0=lc[0], total=total, eps=eps, vp=f32(vp), max_norm=f32(1-lc[0]), all_other=f32(1.0)-total if total<1 else f32(0))
class DeviceModel:
    def __init__(s,n,lc,vp):
        s.p=params(lc,vp); s.n=n; s.lists=[dict() for _ in range(n)]; s.post=np.zeros(n+1,f32); s.was=np.zeros(n,bool); s.empty=True
    def link(s,a,b,m):
        s.lists[a][b]=m; s.lists[b][a]=m
    def update(s, like, inset):
        p=s.p; n=s.n
        cols=1+int(inset.sum())
        pin=np.zeros(n+1,f32); col=[None]*n
        s_in=0.0; s_fill=0.0
        for c in range(n):
            if not inset[c]: continue
            pc=f32(1) if s.empty else (s.post[1+c] if s.was[c] else f32(0))
            sm=f32(0); self_v=f32(0); nz=0; has_self=False
            for r,mg in s.lists[c].items():
                if not inset[r]: continue
                v=p['lc'][mg+1]; sm=f32(sm+v)
                if r==c: has_self=True; self_v=v
                elif v!=0: nz+=1
            delta=f32(0)
            if np.float64(sm) < np.float64(p['total'])-p['lc0']:
                delta=f32(np.float64(p['total'])-p['lc0']-np.float64(sm)); sm=f32(sm+delta)
            if f32(self_v+delta)!=0: nz+=1
            fill=f32(0)
            if p['all_other']>0 and cols>1:
                value=f32(p['all_other']/f32(cols-1)); nzero=(cols-1)-nz
                sm=f32(np.float64(sm)+np.float64(value)*nzero); fill=value
            scale=f32(1); ren=False
            if np.float64(sm)<np.float64(p['max_norm'])-0.0001 or np.float64(sm)>np.float64(p['max_norm'])+0.0001:
                scale=f32(p['max_norm']/sm); ren=True; fill=f32(fill*scale)
                if fill<p['eps']: fill=f32(0)
            col[c]=(scale,delta,fill,ren,has_self); pin[1+c]=pc
            s_in+=float(pc); s_fill+=float(fill)*float(pc)
        pin[0]=f32(1) if s.empty else s.post[0]
        if p['vp']>0:
            if cols>1: vp_col=f32((1.0-np.float64(p['vp']))/(cols-1)); p00=p['vp']
            else: vp_col=f32(0); p00=f32(1)
        elif cols>1: vp_col=f32(1.0/cols); p00=vp_col
        else: vp_col=f32(0); p00=f32(1)
        def fin(v,cs):
            if cs[3]:
                v=f32(v*cs[0])
                if v<p['eps']: v=f32(0)
            return v
        un=np.zeros(n+1,f32); usum=0.0
        for i in range(n):
            if not inset[i]: continue
            acc=0.0; has_self=False
            for c,mg in s.lists[i].items():
                if not inset[c]: continue
                cs=col[c]; v=p['lc'][mg+1]
                if c==i: v=f32(v+cs[1]); has_self=True
                if v==0: continue
                v=fin(v,cs); acc+=(float(v)-float(cs[2]))*float(pin[1+c])
            if not has_self:
                cs=col[i]
                if cs[1]!=0: acc+=(float(fin(cs[1],cs))-float(cs[2]))*float(pin[1+i])
            prior=f32(acc+s_fill+float(vp_col)*float(pin[0]))
            un[1+i]=f32(like[1+i]*prior); usum+=float(un[1+i])
        prior0=f32(float(p00)*float(pin[0])+float(f32(p['lc0']))*s_in)
        un[0]=f32(like[0]*prior0)
        tot=f32(usum+float(un[0]))
        post=np.zeros(n+1,f32)
        for i in range(n+1):
            if i==0 or inset[i-1]: post[i]=f32(un[i]/tot) if tot!=0 else un[i]
        s.post=post; s.was=inset.copy(); s.empty=False
        return post
