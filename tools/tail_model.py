""" Schedule model of the FNN rollout kernel on the episode lengths of one bench launch (tools/dump_lengths.py):
how much of the launch is the drain after the queue has run empty, and what handing the leftover lanes of a wave
to other waves would recover.  Until the queue is empty the waves step in lock-step rounds; the drain is simulated
per SIMD (event driven): a round of a SIMD costs the sum over its waves of E + tiles*TT clocks
(E: the per-wave part of a step, TT: one 16-env tile through the network).
Usage: python tools/tail_model.py [lengths.npy] [donate_at_most ...] [gran=G]
gran=G: a draining wave pays for its busy lanes in units of G lanes (16: whole MFMA tiles; 4 or 8: column groups of the
4x4x4 MFMA form of the network, at G/16 of a tile's cost each). """
import heapq
import sys

import numpy as np

E, TT = 3000., 6825.            # clocks; E + 4*TT = 30 300, the measured issue time of one full wave-step
CUS, WAVES, SIMDS = 256, 12, 4
CLK = 2.4e9

args = sys.argv[1:]
GRAN = next((int(x[5:]) for x in args if x.startswith('gran=')), 16)
args = [x for x in args if not x.startswith('gran=')]
path = args[0] if args and args[0].endswith('.npy') else 'gpurun_out/lengths.npy'
thresholds = [int(x) for x in args if not x.endswith('.npy')] or [0, 16, 32, 48]
ln = np.load(path).astype(np.int64)
n_waves = CUS*WAVES
lanes = n_waves*64

# phase 1: lock-step rounds until the queue is empty
res = ln[:lanes].copy().reshape(n_waves, 64)
nxt = lanes
useful = 0.
t1 = 0.
while nxt < ln.size:
    busy = res > 0
    useful += busy.sum()
    t1 += (WAVES//SIMDS)*(E + 4*TT)
    res[busy] -= 1
    done = busy & (res == 0)
    take = min(int(done.sum()), ln.size - nxt)
    idx = np.flatnonzero(done.reshape(-1))[:take]
    res.reshape(-1)[idx] = ln[nxt:nxt + take]
    nxt += take
useful += res.sum()
ideal = useful*(E + 4*TT)/64/(CUS*SIMDS)
print(f'env-steps {useful:.0f}, ideal {ideal/CLK*1e3:.2f} ms, queue empty at {t1/CLK*1e3:.2f} ms')


def drain(res0, donate_at_most, per_alive=16):
    """ Event-driven drain.  donate_at_most = 0: every wave finishes its own lanes (tiles compacted within the wave). """
    res = [list(r[r > 0]) for r in res0]                 # remaining steps of the busy lanes of each wave
    alive = [True]*n_waves
    n_alive = n_waves
    pool = []
    # wave w = cu*WAVES + k*SIMDS + s runs on SIMD (cu, s)
    simd_waves = [[cu*WAVES + k*SIMDS + s for k in range(WAVES//SIMDS)] for cu in range(CUS) for s in range(SIMDS)]
    heap = [(t1, j) for j in range(CUS*SIMDS)]
    heapq.heapify(heap)
    end = t1
    moved = 0
    while heap:
        t, j = heapq.heappop(heap)
        cost = 0.
        for w in simd_waves[j]:
            if not alive[w]:
                continue
            r = res[w]
            if donate_at_most:
                while pool and len(r) < 64:              # idle lanes take waiting environments
                    r.append(pool.pop())
                if 0 < len(r) <= donate_at_most and n_alive > 1 and len(pool) + len(r) <= per_alive*(n_alive - 1):
                    pool.extend(r)
                    moved += len(r)
                    r.clear()
                    alive[w] = False
                    n_alive -= 1
                    continue
            if not r:
                if not pool:
                    alive[w] = False
                    n_alive -= 1
                continue
            cost += E + -(-len(r)//GRAN)*(TT*GRAN/16.)
            res[w] = [x - 1 for x in r if x > 1]
        if cost > 0. or any(alive[w] for w in simd_waves[j]):
            if cost == 0.:
                cost = 500.                              # polling for environments of other waves
            heapq.heappush(heap, (t + cost, j))
            end = max(end, t + cost)
    return end, moved


for th in thresholds:
    end, moved = drain(res, th)
    print(f'gran {GRAN:2d}, donate when <= {th:2d} busy: total {end/CLK*1e3:7.2f} ms, loss {100*(end/ideal - 1):5.2f} %, '
          f'{moved} environments moved')
