#!/usr/bin/env python3
"""ORACLE-BASED DIAGNOSTIC (test infrastructure: not collected by pytest).  Prices, on the oracle's per-ray octree step counts of the
bench frame (orc_trace_steps), the "straggler hand-off" for the tracer's lock-step casts: cap a wave's cast at K trips, hand the lanes
whose ray is still walking (with their walk state) to a queue, resume them in dense waves of stragglers, repeat for P passes.

    head  = trace_kernel: rounds 0..2 of a pixel's path per 8x8 tile (round 0 = primary ray, uncapped)
    tail  = bounce_kernel: the paths alive at their second hit, compacted 64 to a wave, + the stragglers of every earlier pass

Cost model (wave-instructions): C_STEP per lock-step wave-trip, C_SHADE per round a wave executes, C_RES per wave that resumes
stragglers.  Calibrated against rocprofv3 (profiles/r01: 105 M VALU wave-instructions per frame; this model: 112 M).

Result (menger 1920x1080, 4 bounces, bench camera): the best schedules save 5-6 % of the stage's wave-instructions
    K_head 32, K_tail 16, 3 passes: 0.937 x      K_head 24, K_tail 24, 2 passes: 0.947 x      tail only, K 24: 0.965 x
while the straggler records (96+ bytes each, 130-260 k per frame) double the queue traffic.  Lane utilisation goes from 0.50 to 0.56,
not to the 0.7+ a naive estimate suggests: a straggler's path continues in lock step with other stragglers, whose rays are long-tailed
too, and every extra pass pays its own shading rounds.  Not built."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import sim_schedule as S  # noqa: E402

C_STEP, C_SHADE, C_RES = 100.0, 450.0, 200.0


def run_pass(items_len, items_rest, K):
    """One compacted launch over `items` (remaining trips of the current ray, later rounds of the path), 64 to a wave in order.
    Returns cost, the stragglers it produces, useful lane-trips, wave-trips."""
    n = len(items_len)
    m = max((len(r) for r in items_rest), default=0) + 1
    A = np.zeros((n, m), np.int64)
    A[:, 0] = items_len
    for i, r in enumerate(items_rest):
        A[i, 1:1 + len(r)] = r
    pad = (-n) % 64
    A = np.concatenate([A, np.zeros((pad, m), np.int64)]).reshape(-1, 64, m)
    alive = A[:, :, 0] > 0
    cost, wave_trips, useful, strag = 0.0, 0, 0, []
    for j in range(m):
        L = np.where(alive, A[:, :, j], 0)
        alive &= L > 0
        L = np.where(alive, L, 0)
        mx = L.max(axis=1)
        if K is not None:
            over = alive & (L > K)
            for w, l in zip(*np.nonzero(over)):
                strag.append((L[w, l] - K, A[w, l, j + 1:].copy()))
            useful += np.minimum(L, K).sum()
            alive &= ~over
            mx = np.minimum(mx, K)
        else:
            useful += L.sum()
        wave_trips += mx.sum()
        cost += mx.sum() * C_STEP + (mx > 0).sum() * C_SHADE
    return cost, strag, useful, wave_trips


def trim(r):
    r = np.asarray(r)
    z = np.nonzero(r == 0)[0]
    return r[:z[0]] if len(z) else r


def scheme(rays, K_head, K_tail, passes, split_round=3):
    head = rays[:, :, :split_round]
    cost, wt, useful, strag = 0.0, 0, 0, []
    alive = head[:, :, 0] > 0
    for j in range(split_round):
        L = np.where(alive, head[:, :, j], 0)
        alive &= L > 0
        L = np.where(alive, L, 0)
        mx = L.max(axis=1)
        K = None if j == 0 else K_head
        if K is not None:
            over = alive & (L > K)
            for w, l in zip(*np.nonzero(over)):
                strag.append((L[w, l] - K, trim(rays[w, l, j + 1:])))
            alive &= ~over
            useful += np.minimum(L, K).sum()
            mx = np.minimum(mx, K)
        else:
            useful += L.sum()
        wt += mx.sum()
        cost += mx.sum() * C_STEP + (mx > 0).sum() * C_SHADE
    head_cost = cost
    items = [(rays[w, l, split_round], trim(rays[w, l, split_round + 1:])) for w, l in zip(*np.nonzero(alive & (rays[:, :, split_round] > 0)))] + strag
    tail_cost, npass, nrec = 0.0, 0, len(items)
    for p in range(passes):
        if not items:
            break
        c, s, u, w = run_pass([i[0] for i in items], [i[1] for i in items], K_tail if p < passes - 1 else None)
        tail_cost += c + C_RES * ((len(items) + 63) // 64)
        useful += u
        wt += w
        items = s
        nrec += len(s)
        npass += 1
    return head_cost, tail_cost, useful / (64 * wt), nrec, npass


def main():
    view = sys.argv[1] if len(sys.argv) > 1 else "bench"
    rays = S.tiles_of(S.steps_for(view=view))[:, :, 1:].astype(np.int64)
    print("rays", (rays > 0).sum(), "steps", rays.sum(), "mean", rays.sum() / (rays > 0).sum())
    base = scheme(rays, None, None, 1)
    print("today: head %.1f M + tail %.1f M = %.1f M wave-instructions, lane utilisation %.3f, %d records" %
          (base[0] / 1e6, base[1] / 1e6, (base[0] + base[1]) / 1e6, base[2], base[3]))
    for Kh, Kt, P in [(None, 16, 2), (None, 24, 2), (None, 16, 3), (24, 24, 2), (16, 16, 3), (24, 16, 3), (32, 16, 3), (8, 8, 5)]:
        r = scheme(rays, Kh, Kt, P)
        print(f"K_head {Kh} K_tail {Kt} passes {P}: head {r[0] / 1e6:.1f} M tail {r[1] / 1e6:.1f} M total {(r[0] + r[1]) / 1e6:.1f} M "
              f"({(r[0] + r[1]) / (base[0] + base[1]):.3f} x) utilisation {r[2]:.3f} records {r[3]} passes {r[4]}")


if __name__ == "__main__":
    main()
