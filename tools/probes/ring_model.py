"""Model check of the LDS weight-ring protocol of k_mlp16.hip.hpp (ws_start / ws_sync / ws_advance / ws_refill) under
arbitrary latencies: 8 waves, group 1 (waves 4-7) synchronising half a chunk after group 0 (tune::kStagger), DMA pieces
issued by every wave or by one group only (tune::kDmaGroup), NR fragment registers, RS ring slots.  Every LDS read must see
the chunk it expects.  Pure Python, no GPU: python tools/probes/ring_model.py"""
import bisect
import random
import sys


def run(RS, NR, CF=16, waves=8, stagger=True, dma_grp=-1, chunks=40, seed=0, lat_dma=(50, 4000), lat_rd=(10, 600), t_mfma=(32, 200)):
    rng = random.Random(seed)
    one = dma_grp >= 0
    LPW = CF // 4 if one else CF // waves
    # LDS content: per (slot, frag) sorted list of (time, chunk)
    writes = {}
    def dma_land(t, slot, frag, chunk):
        writes.setdefault((slot, frag), []).append((t, chunk))
    def lds_at(t, slot, frag):
        w = sorted(writes.get((slot, frag), []))
        i = bisect.bisect_right(w, (t, 1 << 60)) - 1
        return w[i][1] if i >= 0 else None
    grp = [(w // 4) if stagger else 0 for w in range(waves)]
    issuer = [(not one) or (w // 4) == dma_grp for w in range(waves)]
    piece0 = [((w & 3) if one else w) * LPW for w in range(waves)]
    clock = [0.0] * waves
    outstanding = [[] for _ in range(waves)]     # completion times, in issue order (monotone)
    def issue(w, chunk, slot):
        for i in range(LPW):
            t = clock[w] + rng.uniform(*lat_dma)
            if outstanding[w]:
                t = max(t, outstanding[w][-1] + 1e-3)      # loads return in order
            outstanding[w].append(t)
            dma_land(t, slot, piece0[w] + i, chunk)
    def wait_vmcnt(w, n):
        while len(outstanding[w]) > n:
            clock[w] = max(clock[w], outstanding[w].pop(0))
        outstanding[w] = [t for t in outstanding[w] if t > clock[w]] if False else outstanding[w]
    # ws_start
    for w in range(waves):
        if issuer[w]:
            for k in range(RS - 1):
                issue(w, k, k)
        wait_vmcnt(w, (RS - 2) * LPW)
    tb = max(clock)
    clock = [tb] * waves
    reads = []      # (time, slot, frag, expected chunk, wave, pos)
    ready = [dict() for _ in range(waves)]      # stream position -> time its fragment arrives in registers
    for w in range(waves):
        for i in range(NR):
            t = clock[w] + rng.uniform(*lat_rd)
            reads.append((t, 0, i, 0, w, -1))
            ready[w][i] = t
    slot_cur = [RS - 1] * waves
    nsync = [0] * waves
    total = chunks * CF
    pos = [0] * waves
    # advance waves in lock-step over barrier intervals: each wave runs until its next sync, then all meet
    done = [False] * waves
    while not all(done):
        for w in range(waves):
            while True:
                if pos[w] >= total:
                    done[w] = True
                    break
                f = pos[w] % CF
                at_sync = (f == 0 and grp[w] == 0) or (f == CF // 2 and grp[w] == 1)
                if at_sync and nsync[w] >= 0 and not getattr(run, "_passed", {}).get((w, pos[w])):
                    break
                run._passed.pop((w, pos[w]), None)
                if f == 0:
                    slot_cur[w] = (slot_cur[w] + 1) % RS
                clock[w] = max(clock[w], ready[w].pop(pos[w]))     # the MFMA needs its fragment (s_waitcnt lgkmcnt)
                clock[w] += rng.uniform(*t_mfma)
                c = pos[w] // CF
                q = f + NR
                t = clock[w] + rng.uniform(*lat_rd)
                ready[w][pos[w] + NR] = t
                if q < CF:
                    reads.append((t, c % RS, q, c, w, pos[w]))
                else:
                    reads.append((t, (c + 1) % RS, q - CF, c + 1, w, pos[w]))
                pos[w] += 1
        if all(done):
            break
        # all live waves are at a sync point (or done): wait + barrier + issue
        live = [w for w in range(waves) if not done[w]]
        for w in live:
            wait_vmcnt(w, (RS - 3) * LPW)
        tb = max(clock[w] for w in live)
        for w in live:
            clock[w] = tb
            k = nsync[w]
            slot = slot_cur[w] if grp[w] == 0 else (slot_cur[w] - 1) % RS
            if issuer[w]:
                issue(w, k + RS - 1, slot)
            nsync[w] += 1
            run._passed[(w, pos[w])] = True
    bad = []
    for t, slot, frag, chunk, w, p in reads:
        got = lds_at(t, slot, frag)
        if got != chunk:
            bad.append((w, p, frag, chunk, got))
    return bad


run._passed = {}

if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 30
    for (RS, NR, st, dg) in [(4, 2, False, -1), (4, 2, True, -1), (4, 4, True, -1), (5, 2, True, -1), (4, 4, True, 0), (4, 8, True, 0), (4, 4, True, 1), (3, 2, True, -1)]:
        worst = 0
        ex = None
        for seed in range(n):
            run._passed = {}
            bad = run(RS, NR, stagger=st, dma_grp=dg, seed=seed)
            if len(bad) > worst:
                worst, ex = len(bad), bad[:3]
        print("RS=%d NR=%d stagger=%s dma_grp=%d: worst stale reads over %d seeds = %d %s" % (RS, NR, st, dg, n, worst, ex or ""))
